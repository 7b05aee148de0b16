#!/usr/bin/env python
"""L2 -> CU fill-rate probe (B/clk/CU) for LDS-DMA vs VGPR loads, as a function of resident blocks per CU."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd  # noqa
from diff_foley_amd import engine as E
L = E.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
buf = torch.randn(64 << 20, device="cuda")       # 256 MB
out = torch.zeros(64, device="cuda")
CLK = 2.4e9


def run(kind, span, stride, n, blocks):
    f = lambda: L.df_test_fill(kind, C.c_void_p(buf.data_ptr()), C.c_void_p(out.data_ptr()), span, stride, n, blocks, st)
    assert f() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3
    byts = span * n * blocks
    return byts / t / 1e12, byts / t / 256 / CLK


for name, span, stride in (("shared 1MB window (L2 hot, every block same lines)", 1 << 20, 0),
                           ("private 64KB windows (L2 hot)", 65536, 65536),
                           ("private 256KB windows (L2/MALL)", 262144, 262144)):
    print(name)
    for kind, kn in ((3, "lds-dma"), (4, "vgpr")):     # kind 5 (vgpr+ds_write) is hoisted by the compiler: not reported
        row = []
        for blocks in (256, 512, 768, 1024):
            n = max(1, (64 << 20) // span)
            tb, bpc = run(kind, span, stride, n, blocks)
            row.append(f"{blocks // 256}blk/CU {tb:5.1f}TB/s {bpc:5.1f}B/clk/CU")
        print(f"  {kn:14s} " + " | ".join(row))
