"""Long race screen under contention: 100 chains of 50 back-to-back launches per kernel family (5000 launches each),
output hashed after every chain.  Families: LDS-DMA GEMM (generic + halo conv), register-staged attention, GroupNorm."""
import ctypes as C, hashlib, os, sys
from collections import Counter
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd  # noqa
from diff_foley_amd import engine as E
L = E.lib("bf16"); st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream); p = lambda t: C.c_void_p(t.data_ptr())
label = sys.argv[1]; chains = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bf = lambda t: t.to(torch.bfloat16)
torch.manual_seed(0)
def run(name, fn, out):
    hs = Counter()
    for _ in range(chains):
        for _ in range(50):
            assert fn() == 0, L.df_last_error()
        torch.cuda.synchronize()
        hs[hashlib.md5(out.view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:8]] += 1
    print(f"{label} {name}: {len(hs)} distinct {sorted(hs.values(), reverse=True)[:5] if len(hs) > 1 else ''}")
M, N, K = 512, 1280, 1280
a, w, c = bf(torch.randn(M, K, device="cuda")), bf(torch.randn(N, K, device="cuda") * .05), torch.empty(M, N, device="cuda")
run("gemm tile3 512x1280x1280", lambda: L.df_test_gemm(p(a), p(w), p(c), M, N, K, 3, 1, st()), c)
run("gemm tile13 sk4", lambda: L.df_test_gemm(p(a), p(w), p(c), M, N, K, 13, 4, st()), c)
NB, H, W, Ci, Co = 4, 16, 64, 128, 128
xa, xw, xb, xc = bf(torch.randn(NB*H*W, Ci, device="cuda")), bf(torch.randn(Co, 9*Ci, device="cuda")*.05), torch.randn(Co, device="cuda"), torch.empty(NB*H*W, Co, device="cuda")
run("halo conv tile5", lambda: L.df_test_conv3x3(p(xa), p(xw), p(xb), p(xc), NB, H, W, Ci, Co, 1, 0, 5, 1, st()), xc)
run("conv tile8 generic", lambda: L.df_test_conv3x3(p(xa), p(xw), p(xb), p(xc), NB, H, W, Ci, Co, 1, 0, 8, 1, st()), xc)
Nn, heads, D, T = 4, 2, 64, 256
q, k, vt, o = bf(torch.randn(Nn, T, 128, device="cuda")), bf(torch.randn(Nn, T, 128, device="cuda")), bf(torch.randn(Nn, 128, T, device="cuda")), torch.empty(Nn, T, 128, dtype=torch.bfloat16, device="cuda")
run("attention", lambda: L.df_test_attention(p(q), 128, p(k), 128, p(vt), T, p(o), 128, Nn, heads, D, T, T, D ** -0.5, st()), o)
x = torch.randn(4, 1024, 64, device="cuda"); g, b = torch.randn(64, device="cuda"), torch.randn(64, device="cuda"); go = torch.empty(4, 1024, 64, dtype=torch.bfloat16, device="cuda")
run("groupnorm", lambda: L.df_test_groupnorm(p(x), 64, 4, 1024, 64, p(g), p(b), 1e-5, 1, p(go), st()), go)
