#!/usr/bin/env python
"""End-to-end timing of one sample() + decode_first_stage() per batch size (not the BASELINE metric; for DESIGN.md)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd as P
from diff_foley_amd import synth

sd = synth.make_state_dict(synth.state_dict_spec(), 0)
m = P.LatentDiffusion(**P.stage2_config())
m.load_state_dict(sd)
m.cuda()
m.autotune(True)
for B in (1, 4, 8):
    feats = synth.synthetic_cavp(B).cuda()
    xT = synth.synthetic_xT(B).cuda()
    c = m.get_learned_conditioning(feats)
    uc = torch.zeros_like(c)
    for name, S in (("DDIM", 25), ("DPM_Solver", 50)):
        for it in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            z, _ = m.sample_log_diff_sampler(c, B, name, S, unconditional_guidance_scale=4.5,
                                             unconditional_conditioning=uc, x_T=xT)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            mel = m.decode_first_stage(z)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        print(f"B={B} {name}-{S}: sample {1e3 * (t1 - t0):8.1f} ms ({S / (t1 - t0):6.1f} steps/s)  "
              f"decode {1e3 * (t2 - t1):7.1f} ms  -> {B / (t2 - t0):6.2f} clips/s")
