#!/usr/bin/env python
"""Run ONE GEMM/conv shape with one tile a few times (for rocprofv3 --pmc passes).
usage: gemm_one.py lin M N K tile sk | conv NB H W Cin Cout tile sk"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd  # noqa
from diff_foley_amd import engine as E
L = E.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
a = sys.argv
if a[1] == "lin":
    M, N, K, tile, sk = map(int, a[2:7])
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") * .02).to(torch.bfloat16)
    Cc = torch.empty(M, N, device="cuda")
    f = lambda: L.df_test_gemm(p(A), p(W), p(Cc), M, N, K, tile, sk, st)
else:
    NB, H, Wd, Cin, Cout, tile, sk = map(int, a[2:9])
    A = torch.randn(NB * H * Wd, Cin, device="cuda").to(torch.bfloat16); W = (torch.randn(Cout, 9 * Cin, device="cuda") * .02).to(torch.bfloat16)
    b = torch.zeros(Cout, device="cuda"); Cc = torch.empty(NB * H * Wd, Cout, device="cuda")
    f = lambda: L.df_test_conv3x3(p(A), p(W), p(b), p(Cc), NB, H, Wd, Cin, Cout, 1, 0, tile, sk, st)
for _ in range(5):
    assert f() == 0
torch.cuda.synchronize()
