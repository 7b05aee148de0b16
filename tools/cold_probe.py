#!/usr/bin/env python
"""Layers of the denoise step with COLD weights (every call takes another of > 700 MB of distinct weight buffers, as in the step,
where 1.7 GB of weights pass through 256 MB of Infinity Cache): us, TFLOP/s and weight-stream rate per (tile, split-K), the
best of the deep-weight-ring tiles (18-20) next to the best of the rest.  usage: cold_probe.py [filter] [tile,tile,...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd  # noqa
from diff_foley_amd import engine as E
from gemm_bench import ptr, TILES

L = E.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
flt = sys.argv[1] if len(sys.argv) > 1 else ""
only = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(TILES)
HALO = (5, 6, 7, 15, 16, 17, 23, 24, 25)
SHAPES = [("conv", 8, 2, 8, 1280, 1280), ("conv", 8, 2, 8, 2560, 1280), ("conv", 8, 4, 16, 1280, 1280), ("conv", 8, 4, 16, 2560, 1280),
          ("conv", 8, 8, 32, 640, 640), ("conv", 8, 16, 64, 320, 320),
          ("lin", 512, 10240, 1280), ("lin", 512, 1280, 6400), ("lin", 512, 3840, 1280), ("lin", 128, 10240, 1280),
          ("lin", 2048, 5120, 640), ("lin", 2048, 640, 3200), ("lin", 8192, 2560, 320), ("lin", 8192, 320, 1600)]
for sh in SHAPES:
    if sh[0] == "conv":
        _, NB, H, W, Cin, Cout = sh
        M, K, N = NB * H * W, 9 * Cin, Cout
        name = f"conv {Cin}->{Cout} @{H}x{W}"
    else:
        _, M, N, K = sh
        name = f"lin {M}x{N}x{K}"
    if flt and flt not in name:
        continue
    wbytes = N * K * 2
    nbuf = max(2, (700 << 20) // wbytes)
    a = torch.randn(M, K if sh[0] == "lin" else sh[4], device="cuda").to(torch.bfloat16)
    ws = [(torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(nbuf)]
    b = torch.zeros(N, device="cuda")
    c = torch.empty(M, N, device="cuda")
    res = []
    for t in only:
        if sh[0] == "lin" and t in HALO:
            continue
        for sk in (1, 2, 3, 4, 6, 8, 12, 16, 32):
            if sk > 1 and K // 64 // sk < 2:
                break
            if sh[0] == "conv":
                call = lambda i: L.df_test_conv3x3(ptr(a), ptr(ws[i % nbuf]), ptr(b), ptr(c), NB, H, W, Cin, Cout, 1, 0, t, sk, st)
            else:
                call = lambda i: L.df_test_gemm(ptr(a), ptr(ws[i % nbuf]), ptr(c), M, N, K, t, sk, st)
            if call(0) != 0:
                continue
            for i in range(3):
                call(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n = max(8, 2 * nbuf)
            for i in range(n):
                call(i)
            e1.record()
            torch.cuda.synchronize()
            res.append((e0.elapsed_time(e1) / n * 1e3, t, sk))
    res.sort()
    fmt = lambda r: f"{TILES[r[1]]}/sk{r[2]}: {r[0]:5.1f}us {2.0 * M * N * K / r[0] / 1e6:5.0f}TF {wbytes / r[0] / 1e6:4.2f}TB/s"
    deep = [r for r in res if r[1] >= 18]
    rest = [r for r in res if r[1] < 18]
    print(f"{name} (M={M}, W {wbytes / 1e6:.1f} MB, cold) | deep-W: " + ", ".join(fmt(r) for r in deep[:3]) + " | other: " +
          ", ".join(fmt(r) for r in rest[:4]), flush=True)
