#!/usr/bin/env python
"""Shader-clock stamps of the persistent GEGLU kernel (ffn.hip DBG build, p.dbg bit 6) on st.ff1 at M = 8192: per block
[entry, prologue issued, first wait begin/end, then per tile: after every K step, after the epilogue]."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd  # noqa
from diff_foley_amd import engine as E
from gemm_bench import ptr

L = E.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 21
extra = int(sys.argv[2]) if len(sys.argv) > 2 else 0
M, K = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (8192, 320)
N1 = 8 * K
a = torch.randn(M, K, device="cuda").to(torch.float16)
w = (torch.randn(N1, K, device="cuda") * 0.05).to(torch.float16)
stats = torch.stack([torch.randn(M, K // 64, device="cuda") * 0.1, torch.rand(M, K // 64, device="cuda") * 64 + 60], dim=-1).contiguous()
cs = torch.randn(N1, device="cuda") * 0.1
bias = torch.randn(N1, device="cuda") * 0.1
out = torch.empty(M, N1 // 2, device="cuda", dtype=torch.float16)
for _ in range(3):
    assert L.df_test_geglu(ptr(a), ptr(w), ptr(stats), ptr(cs), ptr(bias), ptr(out), M, K, N1, tile, 64 | extra | (128 if tile >= 32 else 0), st) == 0
buf = np.zeros(1024 * 32, dtype=np.uint64)
assert L.df_test_scratch_read(buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
buf = buf.reshape(1024, 32)
t0 = buf[:512, 0].min()
for b in (0, 1, 8, 100, 255, 256, 511):
    row = buf[b]
    n = int((row > 0).sum())
    rel = (row[:n] - t0).astype(np.int64)
    print(f"block {b:3d}: start {rel[0]:6d}  " + " ".join(f"{int(d):5d}" for d in np.diff(rel)))
ends = np.array([buf[b][int((buf[b] > 0).sum()) - 1] for b in range(512)]) - t0
print("entry spread (cycles):", int((buf[:512, 0] - t0).max()), " last end:", int(ends.max()), " median end:", int(np.median(ends)))
