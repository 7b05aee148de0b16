#!/usr/bin/env python
"""Probe: UNet in_channels / out_channels other than 4 (the reference's constructor takes any) against the oracle, forward, fused CFG
and a 3-step DDIM sample."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import diff_foley_amd as P
from diff_foley_amd import synth
from oracle import unet as ou
from helpers import rel_l2

for cin, cout in ((1, 1), (3, 3), (4, 8), (8, 4), (8, 8), (9, 4), (16, 16), (4, 1), (4, 3), (4, 64), (64, 4)):
    cfg = dict(synth.UNET_TINY, in_channels=cin, out_channels=cout)
    try:
        sd = synth.make_state_dict(synth.state_dict_spec(cfg, synth.VAE_TINY, synth.COND_TINY), 11)
        m = P.LatentDiffusion(precision="fp16", **dict(P.stage2_config(cfg, synth.VAE_TINY, synth.COND_TINY), channels=cin))
        m.load_state_dict(sd)
        m.cuda()
        usd = ou.sub_state_dict(sd, "model.diffusion_model.")
        g = torch.Generator().manual_seed(cin * 100 + cout)
        x, c, t = torch.randn(2, cin, 16, 32, generator=g), torch.randn(2, 9, 128, generator=g), torch.tensor([700, 3])
        ref = ou.unet_forward(usd, cfg, x, t, c)
        y = m.apply_model(x.cuda(), t.cuda(), c.cuda()).cpu()
        e1 = rel_l2(y, ref)
        uc = torch.zeros_like(c)
        tt = torch.tensor([500, 500])
        e2_ = ou.unet_forward(usd, cfg, torch.cat([x, x]), torch.cat([tt, tt]), torch.cat([uc, c]))
        ref_cfg = e2_[:2] + 3.0 * (e2_[2:] - e2_[:2])
        m.engine.set_context(torch.cat([uc, c]).cuda())
        ycfg = m.engine.unet_forward_cfg(x.cuda(), tt.float().cuda(), 3.0).cpu()
        e2 = rel_l2(ycfg, ref_cfg)
        msg = f"forward {e1:.2e}  cfg {e2:.2e}"
        if cin == cout:
            z, _ = m.sample_log_diff_sampler(c.cuda(), 2, "DDIM", 4, size_len=32, unconditional_guidance_scale=3.0,
                                             unconditional_conditioning=uc.cuda(), x_T=x.cuda())
            msg += f"  sample {tuple(z.shape)} finite={bool(torch.isfinite(z).all())}"
        print(f"in {cin:2d} out {cout:2d}: {msg}")
    except Exception as e:
        print(f"in {cin:2d} out {cout:2d}: RAISED {type(e).__name__}: {str(e)[:160]}")
