#!/usr/bin/env python
"""GroupNorm kernel time on the UNet's shapes (N = 8), back to back, us per launch.  DF_GN_DBG (experiment builds only): 1 = no
stores, 2 = no block reductions."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd  # noqa
from diff_foley_amd import engine as E

L = E.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ptr = lambda t: C.c_void_p(t.data_ptr())
for N, HW, Cc in [(8, 1024, 320), (8, 1024, 640), (8, 1024, 960), (8, 256, 640), (8, 256, 1280), (8, 256, 1920), (8, 64, 1280), (8, 16, 1280)]:
    xs = [torch.randn(N * HW, Cc, device="cuda") for _ in range(12)]           # rotate buffers: inputs come from memory, not L2
    g, b = torch.randn(Cc, device="cuda"), torch.randn(Cc, device="cuda")
    out = torch.empty(N * HW, Cc, device="cuda", dtype=torch.float16)
    def run(i):
        assert L.df_test_groupnorm(ptr(xs[i % 12]), Cc, N, HW, Cc, ptr(g), ptr(b), C.c_float(1e-5), 1, ptr(out), st) == 0
    for i in range(12):
        run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(48):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 48 * 1e3
    mb = N * HW * Cc * 6 / 1e6
    print(f"GN N={N} HW={HW:5d} C={Cc:5d}: {us:6.2f} us  ({mb:5.1f} MB, {mb / us * 1e3:6.0f} GB/s)")
