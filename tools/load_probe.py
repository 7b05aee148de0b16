#!/usr/bin/env python
"""Probe: what the loader does with checkpoints a user may bring -- wrong shapes, missing / extra keys, other dtypes, non-contiguous
tensors, tensors on the device -- and the tune-cache importer with malformed text."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import diff_foley_amd as P
from diff_foley_amd import synth
from helpers import tiny_state_dict


def model(sd, **kw):
    m = P.LatentDiffusion(precision="fp16", **P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY))
    m.load_state_dict(sd, **kw)
    m.cuda()
    return m


def run(m):
    c = m.get_learned_conditioning(synth.synthetic_cavp(1, 32, 64, seed=3).cuda())
    z, _ = m.sample_log_diff_sampler(c, 1, "DDIM", 4, x_T=synth.synthetic_xT(1, seed=4).cuda())
    return m.decode_first_stage(z)


def t(name, f):
    try:
        r = f()
        print(name, "->", r if not torch.is_tensor(r) else tuple(r.shape))
    except Exception as e:
        print(name, "RAISED", type(e).__name__, str(e)[:220])


base = tiny_state_dict()
want = run(model(base))
k_conv = "model.diffusion_model.input_blocks.1.0.in_layers.2.weight"
k_bias = "model.diffusion_model.input_blocks.1.0.in_layers.2.bias"
sd = dict(base); sd[k_conv] = sd[k_conv][:, :-1]
t("conv weight with one input channel less", lambda: run(model(sd)))
sd = dict(base); sd[k_conv] = sd[k_conv].reshape(-1)
t("conv weight flattened", lambda: run(model(sd)))
sd = dict(base); sd[k_bias] = torch.cat([sd[k_bias], sd[k_bias]])
t("bias twice as long", lambda: run(model(sd)))
sd = dict(base); del sd[k_bias]
t("missing bias", lambda: run(model(sd)))
sd = dict(base); sd["model.diffusion_model.not_a_key"] = torch.zeros(3)
t("extra key under the UNet prefix", lambda: bool(torch.equal(run(model(sd)), want)))
sd = {k: v.double() for k, v in base.items()}
t("float64 checkpoint", lambda: bool(torch.equal(run(model(sd)), want)))
sd = {k: v.half().float() for k, v in base.items()}
t("fp16-rounded checkpoint runs", lambda: run(model({k: v.half() for k, v in base.items()})))
t("fp16 storage == its fp32 upcast", lambda: bool(torch.equal(run(model({k: v.half() for k, v in base.items()})), run(model(sd)))))
sd = {k: (v.t().contiguous().t() if v.dim() == 2 else v) for k, v in base.items()}
t("non-contiguous 2-D tensors", lambda: bool(torch.equal(run(model(sd)), want)))
sd = {k: v.cuda() for k, v in base.items()}
t("checkpoint already on the device", lambda: bool(torch.equal(run(model(sd)), want)))
sd = dict(base); sd[k_conv] = torch.full_like(sd[k_conv], float("nan"))
t("NaN weights", lambda: bool(torch.isnan(run(model(sd))).any()))
t("strict=True with an unexpected key", lambda: model(dict(base, junk=torch.zeros(1)), strict=True))
m = model(base)
for text in (b"", b"garbage\n", b"unet 1 2\n", b"\x00\xff\xfe binary \n" * 3, b"a b c d e f g h i j k l m n o p\n" * 1000, b"k 99999 1 0\n"):
    t(f"tune_cache_import({text[:24]!r}...)", lambda: (m.engine.tune_cache_import(text), bool(torch.equal(run(m), want)))[1])
