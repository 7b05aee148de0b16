#!/usr/bin/env python
"""Error budget of the bf16 build, one op family at a time (previous verdict item 6).

The bf16 build's decoded mel is 4.6e-3 MAE away from the fp32 reference, the fp16 build's 6e-4: the difference is operand rounding
(2^-9 against 2^-12 per operand).  WHERE does it come from?  In the fp16 build, df_debug_requant re-rounds the operand-type outputs of
selected op families to bf16 precision right behind the op (activations only: the packed weights stay fp16-rounded); everything else is
untouched.  Workload = the north-star metric: 25-step DDIM, CFG 4.5, B = 1, seed 21, decode_first_stage, mel MAE against the
reference's golden output (tests/golden/g5_full_samplers.npz).  Rows: nothing re-rounded (the fp16 build), every family alone, all
activations, and the real bf16 build (activations AND weights).  usage: tools/error_budget.py > profiles/r6_bf16_error_budget.txt"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd as P
from diff_foley_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = np.load(os.path.join(ROOT, "tests", "golden", "g5_full_samplers.npz"))
ref = g["ddim25_mel_21"]
sd = synth.make_state_dict(synth.state_dict_spec(), 0)


def mel_mae(m):
    xT = synth.synthetic_xT(1, seed=21).cuda()
    c = m.get_learned_conditioning(synth.synthetic_cavp(1, 32, 512, seed=1234).cuda())
    z, _ = m.sample_log_diff_sampler(c, 1, "DDIM", 25, unconditional_guidance_scale=4.5, unconditional_conditioning=torch.zeros_like(c), x_T=xT)
    mel = m.decode_first_stage(z)[:, 0].cpu().numpy()
    zr = float(np.linalg.norm(z.cpu().numpy() - g["ddim25_z_21"]) / np.linalg.norm(g["ddim25_z_21"]))
    return float(np.abs(mel - ref).mean()), zr


FAMILIES = [
    ("(nothing: the fp16 build)", ""),
    ("GroupNorm outputs (61 per step: the conv operands)", "groupnorm"),
    ("ResBlock convs' operand outputs (res.conv1 / conv2 / down / up / conv_in)", "res.,down,up,conv_in"),
    ("st.proj_in + st.attn1.out + st.attn2.out (operand copies of the residual stream)", "st.proj_in,st.attn1.out,st.attn2.out"),
    ("st.qkv (Q | K | V^T)", "st.qkv,st.q2"),
    ("attention outputs", "attn."),
    ("st.xs (cross-attention probabilities) + st.xo", "st.xs,st.xo"),
    ("st.ff1 (GEGLU hidden tensor)", "st.ff1"),
    ("st.ffproj (operand copy of the block output)", "st.ffproj"),
    ("x.pack + time embedding + context operands (ctx.*, t.*)", "x.pack,t.,ctx."),
    ("VAE decoder ops (vae.*, z.pack) [decode only]", "vae.,z.pack"),
    ("ALL UNet activations (every family above but the VAE's)", "groupnorm,res.,down,up,conv_in,st.,attn.,x.pack,t.,ctx.,out.conv"),
    ("ALL activations incl. the VAE decoder", "*"),
    ("all UNet activations EXCEPT the GroupNorm outputs", "res.,down,up,conv_in,st.,attn.,x.pack,t.,ctx.,out.conv"),
    ("all UNet activations EXCEPT GroupNorm outputs, st.xs / st.xo, st.ffproj", "res.,down,up,conv_in,st.proj_in,st.attn,st.q,st.ff1,attn.,x.pack,t.,ctx.,out.conv"),
]

m = P.LatentDiffusion(precision="fp16", **P.stage2_config())
m.load_state_dict(sd)
m.cuda()
print("# tools/error_budget.py: decoded-mel MAE of a 25-step DDIM sample (CFG 4.5, B = 1, seed 21) against the reference's golden output;")
print("# fp16 build with the operand-type OUTPUTS of one op family re-rounded to bf16 precision (df_debug_requant), weights untouched.")
print(f"# mel range {float(ref.max() - ref.min()):.2f}, std {float(ref.std()):.3f}; north-star bound 1e-3")
print(f"{'re-rounded to bf16':85s} {'mel MAE':>10s} {'z rel-L2':>10s} {'MAE - fp16':>11s}")
base = None
for name, pre in FAMILIES:
    m.engine.debug_requant(pre)
    mae, zr = mel_mae(m)
    if base is None:
        base = mae
    print(f"{name:85s} {mae:10.3e} {zr:10.3e} {mae - base:+11.2e}", flush=True)
m.engine.debug_requant("")
del m
# weights only: every checkpoint tensor rounded to bf16 on the host, loaded into the fp16 build (LayerNorm / BatchNorm folds are then
# formed from the rounded factors: close to, not identical with, the bf16 build's rounding of the folded product)
sdw = {k: (v.to(torch.bfloat16).to(v.dtype) if v.is_floating_point() and v.dim() >= 2 else v) for k, v in sd.items()}
mw = P.LatentDiffusion(precision="fp16", **P.stage2_config())
mw.load_state_dict(sdw)
mw.cuda()
mae, zr = mel_mae(mw)
print(f"{'WEIGHT matrices rounded to bf16 on the host, activations fp16':85s} {mae:10.3e} {zr:10.3e} {mae - base:+11.2e}", flush=True)
mw.engine.debug_requant("*")
mae, zr = mel_mae(mw)
print(f"{'  + ALL activations re-rounded (an emulated bf16 build)':85s} {mae:10.3e} {zr:10.3e} {mae - base:+11.2e}", flush=True)
mw.engine.debug_requant("res.,down,up,conv_in,st.proj_in,st.attn,st.q,st.ff1,attn.,x.pack,t.,ctx.,out.conv")
mae, zr = mel_mae(mw)
print(f"{'  + activations EXCEPT GroupNorm outputs, st.xs / st.xo, st.ffproj':85s} {mae:10.3e} {zr:10.3e} {mae - base:+11.2e}", flush=True)
mw.engine.debug_requant("")
del mw
mb = P.LatentDiffusion(precision="bf16", **P.stage2_config())
mb.load_state_dict(sd)
mb.cuda()
mae, zr = mel_mae(mb)
print(f"{'the bf16 build (activations AND weights in bf16)':85s} {mae:10.3e} {zr:10.3e} {mae - base:+11.2e}")
