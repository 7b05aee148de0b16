#!/usr/bin/env python
"""The denoise step against the sampler batch (not the BASELINE metric; for DESIGN.md section 8): ms per CFG step of a 25-step DDIM
``sample_log_diff_sampler`` call on the facade (shipped plan table, no tuning), the step's algorithmic rate (355.72 * B GFLOP per
step, SURVEY.md section 8 d) and its fraction of the dense 16-bit MFMA peak (2.5 PFLOP/s), for B = 1 .. 32.  What does not scale with
B is per-launch latency; what does is the kernels' own rate."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd as P
from diff_foley_amd import synth

BATCHES = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,2,4,8,16,32".split(","))]
sd = synth.make_state_dict(synth.state_dict_spec(), 0)
m = P.LatentDiffusion(**P.stage2_config())
m.load_state_dict(sd)
m.cuda()
print(f"# precision {m.engine.precision}; plans from the shipped table (nearest row count for batches it does not hold)")
print("#  B   ms/step   steps/s   sample-steps/s   TFLOP/s   of MFMA peak   launches")
prev = None
for B in BATCHES:
    feats = synth.synthetic_cavp(B).cuda()
    xT = synth.synthetic_xT(B).cuda()
    c = m.get_learned_conditioning(feats)
    uc = torch.zeros_like(c)
    best = 1e9
    for it in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        z, _ = m.sample_log_diff_sampler(c, B, "DDIM", 25, unconditional_guidance_scale=4.5, unconditional_conditioning=uc, x_T=xT)
        torch.cuda.synchronize()
        if it:
            best = min(best, time.perf_counter() - t0)
    ms = 1e3 * best / 25
    tf = 355.72 * B / ms          # GFLOP / ms = TFLOP/s
    nl = int(m.engine.plan_stats()["launches"])
    print(f"  {B:3d}  {ms:8.3f}  {1e3 / ms:8.1f}  {B * 1e3 / ms:14.1f}  {tf:8.1f}  {tf / 2500:12.3f}  {nl:9d}  finite={bool(torch.isfinite(z).all())}")
    if prev is not None:
        dB, dms = B - prev[0], ms - prev[1]
        print(f"#      marginal: {dms / dB:.3f} ms per added sample = {355.72 * dB / dms:.0f} TFLOP/s ({355.72 * dB / dms / 2500:.3f} of peak) on the added work")
    prev = (B, ms)
