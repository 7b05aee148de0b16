#!/usr/bin/env python
"""Run ONE attention shape a few times (for rocprofv3 --pmc passes / timing).  usage: attn_one.py N heads D Tq Tk [reps]"""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd  # noqa
from diff_foley_amd import engine as E
L = E.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
N, heads, D, Tq, Tk = map(int, sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
Cc = heads * D
q = torch.randn(N, Tq, Cc, device="cuda").to(torch.bfloat16)
k = torch.randn(N, Tk, Cc, device="cuda").to(torch.bfloat16)
ldvt = (Tk + 31) // 32 * 32
vt = torch.randn(N, Cc, ldvt, device="cuda").to(torch.bfloat16)
o = torch.empty(N, Tq, Cc, device="cuda", dtype=torch.bfloat16)
f = lambda: L.df_test_attention(p(q), Cc, p(k), Cc, p(vt), ldvt, p(o), Cc, N, heads, D, Tq, Tk, D ** -0.5, st)
for _ in range(3):
    assert f() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print(f"attention N={N} heads={heads} D={D} Tq={Tq} Tk={Tk}: {us:.1f} us  {4.0 * N * heads * Tq * Tk * D / us / 1e6:.0f} TFLOP/s")
