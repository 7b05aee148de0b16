#!/usr/bin/env python
"""Micro-benchmark of the GroupNorm kernels on the UNet's shapes at N = 8 (df_test_groupnorm through the C ABI).
usage: python tools/gn_bench.py   (DF_LIB_OVERRIDE=<other libdfengine.so> for a same-box A/B)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd  # noqa
from diff_foley_amd import engine as E

SHAPES = [(8, 1024, 320), (8, 1024, 640), (8, 1024, 960), (8, 256, 640), (8, 256, 1280), (8, 256, 1920), (8, 64, 1280),
          (8, 64, 2560), (8, 16, 1280), (8, 16, 2560), (4, 1024, 320), (2, 1024, 320), (1, 1024, 320), (2, 256, 640), (2, 64, 1280), (2, 16, 1280)]


def main():
    L = E.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for N, HW, Cc in SHAPES:
        x = torch.randn(N, HW, Cc, device="cuda")
        g, b = torch.randn(Cc, device="cuda"), torch.randn(Cc, device="cuda")
        out = torch.empty(N, HW, Cc, device="cuda", dtype=torch.bfloat16)
        fn = lambda: L.df_test_groupnorm(C.c_void_p(x.data_ptr()), Cc, N, HW, Cc, C.c_void_p(g.data_ptr()), C.c_void_p(b.data_ptr()),
                                         C.c_float(1e-5), 1, C.c_void_p(out.data_ptr()), st)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        mb = N * HW * Cc * 6 / 1e6
        print(f"GN N={N} HW={HW:5d} C={Cc:5d}: {us:6.2f} us  ({mb:6.1f} MB, {mb / us:5.2f} TB/s incl. the launch)")


if __name__ == "__main__":
    main()
