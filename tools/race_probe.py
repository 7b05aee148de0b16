"""Race screen under GPU contention: every kernel family / tile through the unit-test entry points, many repetitions of
the SAME launch, counting distinct output hashes.  Run two copies at once (see the guide's rule: test hand-offs under
uneven load): a correct kernel gives exactly one hash.  usage: python tools/race_probe.py <label> [reps]"""
import ctypes as C
import hashlib
import os
import sys
from collections import Counter

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd  # noqa: E402,F401
from diff_foley_amd import engine as E  # noqa: E402

L = E.lib("bf16")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
label = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
torch.manual_seed(0)


def screen(name, fn, out):
    hs = Counter()
    for _ in range(reps):
        out.fill_(float("nan")) if out.dtype.is_floating_point else out.zero_()
        rc = fn()
        if rc != 0:
            print(f"{label} {name}: rc {rc} {L.df_last_error()}")
            return
        torch.cuda.synchronize()
        hs[hashlib.md5(out.cpu().numpy().tobytes() if out.dtype != torch.bfloat16 else out.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:8]] += 1
    flag = "" if len(hs) == 1 else "   <-- NONDETERMINISTIC"
    print(f"{label} {name}: {len(hs)} distinct {dict(hs) if len(hs) > 1 else ''}{flag}")


bf = lambda t: t.to(torch.bfloat16)
# generic GEMM tiles, with and without split-K
for (M, N, K) in [(512, 1280, 1280), (2048, 320, 320), (128, 256, 2560)]:
    a, w = bf(torch.randn(M, K, device="cuda")), bf(torch.randn(N, K, device="cuda") * .05)
    c = torch.empty(M, N, device="cuda")
    for tile in (0, 1, 2, 3, 4, 8, 9, 10, 11, 12, 13, 14):
        for sk in (1, 4):
            screen(f"gemm {M}x{N}x{K} tile {tile} sk {sk}", lambda: L.df_test_gemm(p(a), p(w), p(c), M, N, K, tile, sk, st()), c)
# convs (generic implicit GEMM + halo tiles)
for (NB, H, W, Cin, Cout) in [(4, 16, 64, 64, 64), (4, 8, 32, 128, 128), (4, 2, 8, 256, 256)]:
    a = bf(torch.randn(NB * H * W, Cin, device="cuda"))
    w = bf(torch.randn(Cout, 9 * Cin, device="cuda") * .05)
    b = torch.randn(Cout, device="cuda")
    c = torch.empty(NB * H * W, Cout, device="cuda")
    for tile in (0, 3, 5, 6, 7, 8, 13, 15, 16):
        for sk in (1, 2):
            def f():
                rc = L.df_test_conv3x3(p(a), p(w), p(b), p(c), NB, H, W, Cin, Cout, 1, 0, tile, sk, st())
                return 0 if (rc != 0 and b"invalid argument" in L.df_last_error()) else rc
            screen(f"conv {NB}x{H}x{W} {Cin}->{Cout} tile {tile} sk {sk}", f, c)
# attention
for (N, heads, D, Tq, Tk) in [(4, 2, 32, 1024, 1024), (4, 2, 64, 256, 256), (4, 2, 128, 64, 64), (4, 2, 32, 1024, 32)]:
    Cc = heads * D
    q, k = bf(torch.randn(N, Tq, Cc, device="cuda")), bf(torch.randn(N, Tk, Cc, device="cuda"))
    ldvt = (Tk + 31) // 32 * 32
    vt = bf(torch.randn(N, Cc, ldvt, device="cuda"))
    o = torch.empty(N, Tq, Cc, dtype=torch.bfloat16, device="cuda")
    screen(f"attention D={D} Tq={Tq} Tk={Tk}", lambda: L.df_test_attention(p(q), Cc, p(k), Cc, p(vt), ldvt, p(o), Cc, N, heads, D, Tq, Tk, D ** -0.5, st()), o)
# norms
x = torch.randn(4, 1024, 64, device="cuda")
g, b = torch.randn(64, device="cuda"), torch.randn(64, device="cuda")
o = torch.empty(4, 1024, 64, dtype=torch.bfloat16, device="cuda")
screen("groupnorm 4x1024x64", lambda: L.df_test_groupnorm(p(x), 64, 4, 1024, 64, p(g), p(b), 1e-5, 1, p(o), st()), o)
