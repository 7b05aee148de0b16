#!/usr/bin/env python
"""Probe: the FULL model on latent widths other than 64 (size_len of sample_log_diff_sampler: 4 s / 16 s / 24 s of audio) -- one
forward against the oracle's fp32 forward on the same inputs, the shipped-table plans (nearest row count) and the cost-model plans,
plus VAE decode of the same width."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import diff_foley_amd as P
from diff_foley_amd import synth
from oracle import unet as ou, vae as ov
from helpers import rel_l2

sd = synth.make_state_dict(synth.state_dict_spec(), 0)
usd = ou.sub_state_dict(sd, "model.diffusion_model.")
vsd = ou.sub_state_dict(sd, "first_stage_model.")
m = P.LatentDiffusion(**P.stage2_config())
m.load_state_dict(sd)
m.cuda()
torch.set_num_threads(64)
for W in [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["32", "128"])]:
    g = torch.Generator().manual_seed(W)
    x, c, t = torch.randn(1, 4, 16, W, generator=g), torch.randn(1, 32, 768, generator=g) * 0.05, torch.tensor([481])
    t0 = time.time()
    ref = ou.unet_forward(usd, synth.UNET_FULL, x, t, c)
    t1 = time.time()
    y = m.apply_model(x.cuda(), t.cuda(), c.cuda()).cpu()
    st = m.engine.plan_stats()
    uc = torch.zeros_like(c)
    m.engine.set_context(torch.cat([uc, c]).cuda())
    ycfg = m.engine.unet_forward_cfg(x.cuda(), t.float().cuda(), 1.0).cpu()       # scale 1: the cond half
    m._ctx_owner = None
    z = torch.randn(1, 4, 16, W, generator=g)
    dref = ov.decode_first_stage(vsd, synth.VAE_FULL, z)
    d = m.decode_first_stage(z.cuda()).cpu()
    print(f"W={W}: unet rel-L2 {rel_l2(y, ref):.2e} (cfg-plan cond half {rel_l2(ycfg, ref):.2e}), {st['launches']} launches; "
          f"vae rel-L2 {rel_l2(d, dref):.2e}  [oracle forward {t1 - t0:.0f} s]", flush=True)
