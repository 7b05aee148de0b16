#!/usr/bin/env python
"""The LayerNorm-folded GEGLU projection (st.ff1) in isolation on its three shapes: generic tiles vs the persistent kernel (tiles
21 / 22), the latter also with its debug switches (no MFMA / no stores / no GELU / no operand requests), weights cold (cycled)."""
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
TILES = dict(TILES)
TILES.update({21: "P128", 22: "P64", 30: "P128w8", 31: "P128w8L", 32: "W256", 33: "W128", 34: "W64"})
PG = (21, 22, 30, 31, 32, 33, 34)
SWITCHES = [0, 8, 6, 15, 31, 47, 63] if len(sys.argv) > 2 else [0]      # second argument: also the debug switches
tiles = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8, 9, 10, 11, 12, 13, 18, 21, 22, 30, 31, 32, 33, 34]
for (M, K) in ((8192, 320), (2048, 640), (512, 1280)):
    N1 = 8 * K
    nbuf = 1 if os.environ.get('PROBE_WARM') else max(2, (600 << 20) // (N1 * K * 2))
    a = torch.randn(M, K, device="cuda").to(torch.float16)
    ws = [(torch.randn(N1, K, device="cuda") * 0.05).to(torch.float16) for _ in range(nbuf)]
    stats = torch.stack([torch.randn(M, K // 64, device="cuda") * 0.1, torch.rand(M, K // 64, device="cuda") * 64 + 60], dim=-1).contiguous()
    cs = torch.randn(N1, device="cuda") * 0.1
    bias = torch.randn(N1, device="cuda") * 0.1
    out = torch.empty(M, N1 // 2, device="cuda", dtype=torch.float16)
    line = []
    for t in tiles:
        for dbg in ([0] if t not in PG else SWITCHES):
            wide = t in (32, 33, 34)      # wide tiles: ONE weight buffer (the test entry keeps its 320-column packing: dbg bit 7), warm
            call = lambda i: L.df_test_geglu(ptr(a), ptr(ws[0 if wide else i % nbuf]), ptr(stats), ptr(cs), ptr(bias), ptr(out), M, K, N1, t,
                                             dbg | (128 if wide else 0), st)
            if call(0) != 0:
                continue
            for i in range(3):
                call(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n = max(10, 2 * nbuf)
            for i in range(n):
                call(i)
            e1.record()
            torch.cuda.synchronize()
            line.append(f"{TILES[t]}{'/d%d' % dbg if dbg else ''}: {e0.elapsed_time(e1) / n * 1e3:5.1f}")
    print(f"ff1 {M}x{N1}x{K}: " + "  ".join(line), flush=True)
