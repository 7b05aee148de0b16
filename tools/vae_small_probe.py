#!/usr/bin/env python
"""Probe behind tests/test_vae_cond_cavp_fuzz_gpu.py (round 6): VAE decoder parity against the oracle over latent sizes / batches for
one decoder configuration -- localises a shape-dependent difference."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import diff_foley_amd as P
from diff_foley_amd import synth
from oracle import unet as ou, vae as ov
from helpers import rel_l2

for cfg in (dict(z_channels=4, embed_dim=4, ch=128, ch_mult=[1, 2], num_res_blocks=1, out_ch=1),
            dict(z_channels=4, embed_dim=4, ch=64, ch_mult=[1, 2], num_res_blocks=1, out_ch=1),
            dict(z_channels=4, embed_dim=4, ch=128, ch_mult=[1, 2], num_res_blocks=1, out_ch=3)):
    sd = synth.make_state_dict(synth.state_dict_spec(synth.UNET_TINY, cfg, synth.COND_TINY), 705)
    vsd = ou.sub_state_dict(sd, "first_stage_model.")
    for prec in ("fp16", "bf16"):
        m = P.LatentDiffusion(precision=prec, **P.stage2_config(synth.UNET_TINY, cfg, synth.COND_TINY))
        m.load_state_dict(sd)
        m.cuda()
        for (H, W) in ((4, 8), (4, 16), (8, 8), (8, 16), (2, 8), (16, 16)):
            for B in (1, 3):
                z = torch.randn(B, 4, H, W, generator=torch.Generator().manual_seed(805))
                ref = ov.decode_first_stage(vsd, cfg, z)
                y = m.decode_first_stage(z.cuda()).cpu()
                per = [rel_l2(y[i], ref[i]) for i in range(B)]
                print(f"ch={cfg['ch']} out_ch={cfg['out_ch']} {prec} {H}x{W} B={B}: rel-L2 {rel_l2(y, ref):.2e}  per sample {[f'{e:.1e}' for e in per]}")
