"""Times the small-M weight-streaming linears of the time-embedding path (csrc/elementwise.hip) through the C ABI:
register variant vs LDS-staged variant, HIP events on the launch stream.  usage: python tools/linrows_bench.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd  # noqa: E402,F401
from diff_foley_amd import engine as E  # noqa: E402

L = E.lib("bf16")
ptr = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K) in [(8, 20160, 1280), (8, 1280, 1280), (8, 1280, 320), (16, 20160, 1280), (8, 64, 1280)]:
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    flush = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    for variant in (0, 1):
        ts = []
        for it in range(12):
            flush.zero_()                       # weights cold (MALL is 256 MB), as in the step loop
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = L.df_test_linear_rows(ptr(a), K, None, 0, ptr(w), ptr(b), ptr(out), N, M, N, K, 0, variant, st())
            e1.record()
            assert rc == 0, L.df_last_error()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = sorted(ts[2:])
        print(f"M={M} N={N} K={K} variant={variant}: median {ts[len(ts)//2]:.1f} us  min {ts[0]:.1f} us  "
              f"({N*K*2/ts[len(ts)//2]/1e6:.2f} TB/s weights)")
