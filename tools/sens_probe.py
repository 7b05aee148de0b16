#!/usr/bin/env python
"""Sensitivity of a tiny bf16 DDIM run to rounding-level changes: saves the latent of a 6-step CFG run (argv[1] = output file);
compare two processes that differ by an env switch (DF_NO_COLSTATS, DF_TILE_SKIP ...)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import diff_foley_amd as P
from diff_foley_amd import synth
from helpers import tiny_state_dict

prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
m = P.LatentDiffusion(precision=prec, **P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY))
m.load_state_dict(tiny_state_dict())
m.cuda()
B = 2
x = synth.synthetic_xT(B, seed=5).cuda()
c = m.get_learned_conditioning(synth.synthetic_cavp(B, 32, 64, seed=1234).cuda())
z, _ = m.sample_log_diff_sampler(c, B, "DDIM", 6, x_T=x.clone(), unconditional_guidance_scale=4.5, unconditional_conditioning=torch.zeros_like(c))
e1 = m.engine.unet_forward_cfg(x, torch.full((B,), 961.0, device="cuda"), 4.5)
torch.save({"z": z.cpu(), "e1": e1.cpu()}, sys.argv[1])
