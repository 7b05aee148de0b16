#!/usr/bin/env python
"""Weight-streaming 3x3 convs of the 8x-downsampled levels (M = 128 / 512 rows, 29.5 - 59 MB of weights per conv) with COLD
weights (cycled through > 256 MB of distinct buffers, as in the denoise step): us and weight-stream rate per (tile, split-K)."""
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
only = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else list(TILES)
for (NB, H, W, Cin, Cout) in ((8, 2, 8, 1280, 1280), (8, 2, 8, 2560, 1280), (8, 4, 16, 1280, 1280)):
    M, K = NB * H * W, 9 * Cin
    wbytes = Cout * K * 2
    nbuf = max(2, (700 << 20) // wbytes)
    a = torch.randn(M, Cin, device="cuda").to(torch.bfloat16)
    ws = [(torch.randn(Cout, K, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(nbuf)]
    b = torch.zeros(Cout, device="cuda")
    c = torch.empty(M, Cout, device="cuda")
    res = []
    for t in only:
        for sk in (1, 2, 4, 8, 16, 32):
            call = lambda i: L.df_test_conv3x3(ptr(a), ptr(ws[i % nbuf]), ptr(b), ptr(c), NB, H, W, Cin, Cout, 1, 0, t, sk, st)
            if call(0) != 0:
                continue
            for i in range(5):
                call(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n = 3 * nbuf
            for i in range(n):
                call(i)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / n * 1e3
            res.append((us, t, sk))
    res.sort()
    print(f"conv {Cin}->{Cout} @{H}x{W} (M={M}, W {wbytes / 1e6:.1f} MB, cold): " +
          ", ".join(f"{TILES[t]}/sk{sk}: {us:5.1f}us {wbytes / us / 1e6:4.2f}TB/s" for us, t, sk in res[:8]), flush=True)
