#!/usr/bin/env python
"""How long is one K step of the small-tile GEMM?  Times M x N x K for growing K on one tile config (split-K 1): the slope is
the per-64-wide-K-step cost of a latency-bound block, the intercept the launch + fill + epilogue."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd  # noqa
from diff_foley_amd import engine as E
from gemm_bench import timeit, ptr


def main():
    L = E.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for (M, N) in ((512, 1280), (2048, 640), (128, 1280), (8192, 320)):
        for tile in (3, 13, 1, 0):
            row = []
            for K in (64, 128, 320, 640, 1280, 2560, 5120):
                a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
                w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
                c = torch.empty(M, N, device="cuda")
                rc = L.df_test_gemm(ptr(a), ptr(w), ptr(c), M, N, K, tile, 1, st)
                if rc != 0:
                    row.append("   -  ")
                    continue
                us = timeit(lambda: L.df_test_gemm(ptr(a), ptr(w), ptr(c), M, N, K, tile, 1, st), 50)
                row.append(f"{us:6.1f}")
            print(f"M={M:5d} N={N:5d} tile {tile:2d}: K=64..5120 us: " + " ".join(row), flush=True)


main()
