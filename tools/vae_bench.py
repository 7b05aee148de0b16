#!/usr/bin/env python
"""decode_first_stage at B=4 a few times (for rocprofv3 --kernel-trace --stats and the --pmc traffic passes of the
VAE decoder; BASELINE.json configs[1] decodes 4 clips per sample() call).  usage: python tools/vae_bench.py [reps] [per-op csv]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd as P
from diff_foley_amd import synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
m = P.LatentDiffusion(**P.stage2_config())
m.load_state_dict(synth.make_state_dict(synth.state_dict_spec(), 0))
m.cuda()
z = synth.synthetic_xT(4).cuda()
for _ in range(3):
    m.decode_first_stage(z)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()                # wall clock around a synchronised batch (the first torch timing event of a process
for _ in range(reps):                   # was seen to halve the speed of the launches it brackets)
    m.decode_first_stage(z)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3 / reps
if len(sys.argv) > 2:                   # per-op CSV of one instrumented decode (HIP events around every op)
    m.engine.profile_begin()
    m.decode_first_stage(z)
    torch.cuda.synchronize()
    m.engine.profile_end()
    m.engine.profile_dump(sys.argv[2])
print(f"vae decode B=4: {ms:.3f} ms  ({622.2 * 4 / ms:.1f} TFLOP/s algorithmic, {(0.0989 + 0.6096 * 4) / ms * 1e3:.0f} GB/s algorithmic)")
