#!/usr/bin/env python
"""tools/gemm_stamps.py with the launch made COLD, as it is in the plan: before the stamped launch a 1 GiB copy sweeps L2 and the
Infinity Cache, another GEMM shape runs (other code in the instruction caches), and fresh operand tensors are used.  Prints the same
stamp columns for the warm third-in-a-row launch and for the cold one.  usage: gemm_stamps_cold.py tile M N K"""
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


def operands():
    return (torch.randn(M, K, device="cuda").to(torch.float16), (torch.randn(N, K, device="cuda") * 0.02).to(torch.float16),
            torch.empty(M, N, device="cuda"))


def stamps(tag):
    buf = np.zeros(4096 * 32, dtype=np.uint64)
    assert L.df_test_scratch_read(buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
    buf = buf.reshape(4096, 32)
    nb = int((buf[:, 0] > 0).sum())
    tot, d0, d1, d2, dk = [], [], [], [], []
    for blk in range(nb):
        row = buf[blk].astype(np.int64)
        n = int((row[:24] > 0).sum())
        last = max(int(row[24:28].max()), int(row[n - 1]))
        tot.append(last - int(row[0]))
        dd = np.diff(row[:n])
        d0.append(dd[0]); d1.append(dd[1]); d2.append(dd[2])
        if row[31] > 0:
            dk.append(int(row[0]) - int(row[31]))      # first instruction of the wavefront -> first stamp (= the first kernel-argument wait)
    first = min(int(buf[b][0]) for b in range(nb))
    end = max(max(int(buf[b][24:28].max()), int(buf[b][:24].max())) for b in range(nb))
    print(f"{tag:5s} blocks {nb}: wave start->args {int(np.median(dk)) if dk else -1:5d}  entry->requests {int(np.median(d0)):5d}  ->first tile {int(np.median(d1)):5d}  K loop {int(np.median(d2)):6d}  "
          f"block lifetime median {int(np.median(tot)):6d} max {int(max(tot)):6d}  (first entry -> last stamp, one XCD clock: {end - first} cycles)")


a, w, c = operands()
for _ in range(3):
    assert L.df_test_gemm(ptr(a), ptr(w), ptr(c), M, N, K, tile, 1, st) == 0
stamps("warm")
big = torch.empty(256 << 20, device="cuda", dtype=torch.float32)
oa, ow, oc = torch.randn(1024, 640, device="cuda").half(), torch.randn(640, 640, device="cuda").half(), torch.empty(1024, 640, device="cuda")
for rep in range(3):
    a, w, c = operands()
    torch.cuda.synchronize()
    big.add_(1.0)                                                  # 2 GiB of traffic: L2 and the Infinity Cache hold nothing of ours
    assert L.df_test_gemm(ptr(oa), ptr(ow), ptr(oc), 1024, 640, 640, 3, 1, st) == 0      # another kernel's code in the instruction caches
    torch.cuda.synchronize()
    buf0 = np.zeros(4096 * 32, dtype=np.uint64)
    assert L.df_test_gemm(ptr(a), ptr(w), ptr(c), M, N, K, tile, 1, st) == 0
    stamps("cold")
