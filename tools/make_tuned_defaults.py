#!/usr/bin/env python
"""Generates the shipped plan tables diff_foley_amd/tuned/<arch>_<CUs>cu_<precision>.txt (engine.load_tuned_defaults): runs the
in-plan autotuner on THIS GPU for the shapes BASELINE.json's configs use and their neighbours -- UNet denoise steps at sampler batch
1-8 and 16 (with and without CFG batch duplication, hoisted and in-step time embedding), the condition encoder, VAE decode at the same
batches, the double-guidance classifier (forward + gradient) at batch 1 / 2 / 4 / 8, and the on-device CAVP encoder of configs[4] -- and writes the
autotuner's choices (df_tune_cache_export) next to the package.  Run once per GPU model and precision; commit the text files.
usage: tools/make_tuned_defaults.py [bf16|fp16 ...]"""
import os
import sys
import time

os.environ["DF_TUNED_DEFAULTS"] = "0"           # tune from the cost-model plans, not from an older table
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd as P
from diff_foley_amd import engine as E, synth


def tune(precision):
    t0 = time.perf_counter()
    sd = synth.make_state_dict(synth.state_dict_spec(), 0)
    m = P.LatentDiffusion(precision=precision, **P.stage2_config())
    m.load_state_dict(sd)
    m.cuda()
    m.autotune(True)
    eng = m.engine
    for B in (4, 8, 1, 2, 3, 5, 6, 7, 16):
        feats = synth.synthetic_cavp(B).cuda()
        xT = synth.synthetic_xT(B).cuda()
        c = m.get_learned_conditioning(feats)
        uc = torch.zeros_like(c)
        # the reference-shaped entry points build the plans a user's call builds (CFG + hoisted time embedding, then no CFG)
        z, _ = m.sample_log_diff_sampler(c, B, "DDIM", 2, unconditional_guidance_scale=4.5, unconditional_conditioning=uc, x_T=xT)
        m.sample_log_diff_sampler(c, B, "DDIM", 2, unconditional_guidance_scale=1.0, x_T=xT)
        eng.set_context(torch.cat([uc, c]))
        eng.unet_forward_cfg(xT, torch.full((B,), 500.0, device="cuda"), 4.5)        # in-step time embedding form
        m.decode_first_stage(z)
        torch.cuda.synchronize()
        print(f"[{precision}] B={B} tuned, {time.perf_counter() - t0:.0f} s", flush=True)
    cls = P.AlignmentClassifier(classifier_config=dict(params=dict(synth.CLS_FULL)))
    cls.load_state_dict(synth.make_state_dict(synth.classifier_spec(synth.CLS_FULL), 0))
    cls.attach(m)
    for B in (8, 4, 2, 1):          # configs[2] runs 8; the notebook 4 candidates per window
        feats = synth.synthetic_cavp(B, 33).cuda()
        xT = synth.synthetic_xT(B).cuda()
        c = m.get_learned_conditioning(feats[:, :32])
        m.sample_log_with_classifier_diff_sampler(c, origin_cond=feats, batch_size=B, sampler_name="DPM_Solver", ddim_steps=3,
                                                  unconditional_guidance_scale=4.5, unconditional_conditioning=torch.zeros_like(c),
                                                  classifier=cls, classifier_guide_scale=50.0, x_T=xT)
        torch.cuda.synchronize()
    print(f"[{precision}] classifier tuned, {time.perf_counter() - t0:.0f} s", flush=True)
    # BASELINE configs[4]: the on-device CAVP video encoder (one 8 s clip = 32 frames at 224 x 224)
    cavp = P.CAVPInference(embed_dim=512, precision=precision)
    cavp.load_state_dict(synth.make_state_dict(synth.cavp_spec(), 0))
    cavp.cuda()
    cavp.autotune(True)
    cavp.encode_video(synth.synthetic_video(1, 32, 224).cuda(), normalize=True, pool=False)
    torch.cuda.synchronize()
    print(f"[{precision}] CAVP tuned, {time.perf_counter() - t0:.0f} s", flush=True)
    text = eng.tune_cache_export()
    pr = torch.cuda.get_device_properties(0)
    arch = pr.gcnArchName.split(":")[0]
    path = os.path.join(E.TUNED_DIR, f"{arch}_{pr.multi_processor_count}cu_{precision}.txt")
    os.makedirs(E.TUNED_DIR, exist_ok=True)
    with open(path, "wb") as f:
        f.write(text)
    out = os.path.join(os.path.dirname(E.TUNED_DIR), "..", "gpurun_out", os.path.basename(path))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "wb") as f:
        f.write(text)
    n_lines = text.count(b"\n")
    print(f"[{precision}] {n_lines} entries -> {path} ({pr.name})", flush=True)
    del cls, m, cavp


if __name__ == "__main__":
    for p in (sys.argv[1:] or ["fp16", "bf16"]):
        tune(p)
