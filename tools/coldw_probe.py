#!/usr/bin/env python
"""How much slower is a GEMM whose weights come cold from HBM (as in the denoise step: 1.7 GB of weights per step cycle
through the 256 MB MALL) than the same GEMM with MALL/L2-warm weights (what a micro-benchmark loop measures)?
Cycles through enough distinct weight buffers to exceed the MALL."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd  # noqa
from diff_foley_amd import engine as E
from gemm_bench import ptr

L = E.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K, tiles) in ((512, 1280, 1280, (3, 13, 1)), (512, 10240, 1280, (3, 0, 8)), (2048, 5120, 640, (8, 11, 3)), (128, 1280, 1280, (3,))):
    wbytes = N * K * 2
    nbuf = max(2, min(400, (600 << 20) // wbytes))
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    ws = [(torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(nbuf)]
    c = torch.empty(M, N, device="cuda")
    for tile in tiles:
        def run(cold, iters=200):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for i in range(10):
                L.df_test_gemm(ptr(a), ptr(ws[i % nbuf if cold else 0]), ptr(c), M, N, K, tile, 1, st)
            torch.cuda.synchronize()
            e0.record()
            for i in range(iters):
                L.df_test_gemm(ptr(a), ptr(ws[i % nbuf if cold else 0]), ptr(c), M, N, K, tile, 1, st)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters * 1e3
        w, cd = run(False), run(True)
        print(f"M={M} N={N} K={K} tile {tile}: W {wbytes / 1e6:.1f} MB x {nbuf} buffers: warm {w:6.1f} us  cold {cd:6.1f} us  ({cd / w:.2f}x)", flush=True)
