#!/usr/bin/env python
"""Shader-clock stamps of the halo conv kernel (DF_GEMM_DBG=64, split-K 1): per block [entry, prologue issued, first wait begin /
end, after every 64-channel slice (9 taps), after the drain, after the epilogue].  usage: halo_stamps.py tile NB H W Cin Cout"""
import ctypes as C
import os
import sys

import numpy as np
import torch

os.environ["DF_GEMM_DBG"] = "64"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd  # noqa
from diff_foley_amd import engine as E
from gemm_bench import ptr

L = E.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
tile, NB, H, W, Cin, Cout = [int(x) for x in sys.argv[1:7]]
a = torch.randn(NB * H * W, Cin, device="cuda").to(torch.float16)
w = (torch.randn(Cout, 9 * Cin, device="cuda") * 0.02).to(torch.float16)
b = torch.zeros(Cout, device="cuda")
c = torch.empty(NB * H * W, Cout, device="cuda")
for _ in range(3):
    assert L.df_test_conv3x3(ptr(a), ptr(w), ptr(b), ptr(c), NB, H, W, Cin, Cout, 1, 0, tile, 1, st) == 0
buf = np.zeros(4096 * 32, dtype=np.uint64)
assert L.df_test_scratch_read(buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
buf = buf.reshape(4096, 32)
nb = int((buf[:, 0] > 0).sum())
print("blocks:", nb)
# (the clocks of different XCDs are not aligned: only differences inside one block mean anything)
tot = np.array([buf[i, :24][buf[i, :24] > 0][-1] - buf[i, 0] for i in range(nb)], dtype=np.int64)
from gemm_bench import timeit
us = timeit(lambda: L.df_test_conv3x3(ptr(a), ptr(w), ptr(b), ptr(c), NB, H, W, Cin, Cout, 1, 0, tile, 1, st))
print(f"block totals min / median / max {int(tot.min())} / {int(np.median(tot))} / {int(tot.max())} cycles; {us:.1f} us per launch back to back "
      f"(stamps on)")
for blk in (0, 1, 8, 100, nb - 1):
    row = buf[blk].astype(np.int64)
    n = int((row[:24] > 0).sum())
    epi = " | epilogue_block: barrier+park %d, FiLM/residual pre-add %d, store loop %d" % (
        int(row[25] - row[24]), int(row[26] - row[25]), int(row[n - 1] - row[26])) if row[26] > 0 else ""
    print(f"block {blk:3d}: " + " ".join(f"{int(d):6d}" for d in np.diff(row[:n])), " total", int(row[n - 1] - row[0]), epi)
