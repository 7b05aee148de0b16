#!/usr/bin/env python
"""Shader-clock stamps of the generic GEMM kernel (DF_GEMM_DBG=64, split-K 1): per block [entry -> prologue requests issued -> first
tile landed -> K loop done | epilogue: barrier + park, pre-add, store loop].  usage: gemm_stamps.py tile M N K"""
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
tile, M, N, K = [int(x) for x in sys.argv[1:5]]
a = torch.randn(M, K, device="cuda").to(torch.float16)
w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.float16)
c = torch.empty(M, N, device="cuda")
for _ in range(3):
    assert L.df_test_gemm(ptr(a), ptr(w), ptr(c), M, N, K, tile, 1, st) == 0
buf = np.zeros(4096 * 32, dtype=np.uint64)
assert L.df_test_scratch_read(buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
buf = buf.reshape(4096, 32)
nb = int((buf[:, 0] > 0).sum())
print(f"tile {tile} {M}x{N}x{K}: blocks {nb}")
tot = []
for blk in range(nb):
    row = buf[blk].astype(np.int64)
    n = int((row[:24] > 0).sum())
    last = max(int(row[24:28].max()), int(row[n - 1]))
    tot.append(last - int(row[0]))
    if blk in (0, 1, 8, nb // 2, nb - 1):
        epi = " | epilogue stamps (rel. to loop end): " + " ".join(str(int(x - row[n - 1])) for x in row[24:27] if x > 0)
        print(f"block {blk:4d}: " + " ".join(f"{int(d):6d}" for d in np.diff(row[:n])), epi)
print("block lifetime to the last stamp (cycles): median", int(np.median(tot)), "max", int(max(tot)))
