"""Probe: how much of a small GEMM's ~10 us is fixed cost that depends on code size / instruction-cache state?
Times one small GEMM (a) back to back with itself (warm instruction cache) and (b) alternating with other kernels
(cold, as inside the UNet plan), for whichever libdfengine build DF_LIB_OVERRIDE selects."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd  # noqa: E402,F401
from diff_foley_amd import engine as E  # noqa: E402

L = E.lib("bf16")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())


def mk(M, N, K):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * .02).to(torch.bfloat16)
    return A, W, torch.empty(M, N, device="cuda")


def timed(fns, iters=40):
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        for f in fns:
            f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (M, N, K, tile) in [(512, 1280, 1280, 3), (8192, 320, 320, 13), (128, 1280, 1280, 3), (64, 64, 64, 3)]:
    a = mk(M, N, K)
    g = lambda: L.df_test_gemm(p(a[0]), p(a[1]), p(a[2]), M, N, K, tile, 1, st)
    others = []
    for (m2, n2, k2, t2) in [(2048, 640, 640, 0), (512, 1280, 1280, 1), (2048, 1280, 640, 8), (512, 1280, 640, 2), (1024, 640, 640, 9)]:
        b = mk(m2, n2, k2)
        others.append((lambda b=b, m2=m2, n2=n2, k2=k2, t2=t2: L.df_test_gemm(p(b[0]), p(b[1]), p(b[2]), m2, n2, k2, t2, 1, st)))
    x = torch.randn(8, 1024, 320, device="cuda")
    gam, bet = torch.ones(320, device="cuda"), torch.zeros(320, device="cuda")
    o = torch.empty(8, 1024, 320, dtype=torch.bfloat16, device="cuda")
    others.append(lambda: L.df_test_groupnorm(p(x), 320, 8, 1024, 320, p(gam), p(bet), 1e-5, 1, p(o), st))
    warm = timed([g])
    t_oth = timed(others)
    mix = timed([f for o_ in others for f in (g, o_)])
    cold = (mix - t_oth) / len(others)
    print(f"GEMM {M}x{N}x{K} tile {tile}: warm {warm:.2f} us   cold (interleaved with {len(others)} other kernels) {cold:.2f} us")
