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
m.cuda()          # (round 6: no autotune call -- the facade runs the shipped plan table, like a notebook user)
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

# ---- BASELINE.json configs[2]: B = 8, 50-step DPM-Solver++ with the double-guidance classifier in the loop
cls = P.AlignmentClassifier(classifier_config=dict(params=dict(synth.CLS_FULL)))
cls.load_state_dict(synth.make_state_dict(synth.classifier_spec(synth.CLS_FULL), 0))
cls.attach(m)
B = 8
feats = synth.synthetic_cavp(B, 33).cuda()
xT = synth.synthetic_xT(B).cuda()
c = m.get_learned_conditioning(feats[:, :32])
uc = torch.zeros_like(c)
for it in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    z, _ = m.sample_log_with_classifier_diff_sampler(c, origin_cond=feats, batch_size=B, sampler_name="DPM_Solver",
                                                     ddim_steps=50, unconditional_guidance_scale=4.5,
                                                     unconditional_conditioning=uc, classifier=cls,
                                                     classifier_guide_scale=50.0, x_T=xT)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
print(f"config[2] B=8 DPM-Solver++-50 + classifier guidance: {1e3 * (t1 - t0):8.1f} ms ({50 / (t1 - t0):6.1f} steps/s), "
      f"finite={bool(torch.isfinite(z).all())}")

# ---- BASELINE.json configs[4] (per GPU): on-device CAVP encoder -> cond stage -> sampler -> VAE decode, fp16 operands,
# one 8 s video (32 frames at 4 fps, 224x224) x 8 candidates
del m, cls
torch.cuda.empty_cache()
for prec in ("fp16", "bf16"):
    cavp = P.CAVPInference(embed_dim=512, precision=prec)
    cavp.load_state_dict(synth.make_state_dict(synth.cavp_spec(), 0))
    cavp.cuda()
    m = P.LatentDiffusion(precision=prec, **P.stage2_config())
    m.load_state_dict(sd)
    m.cuda()
    video = synth.synthetic_video(1, 32, 224).cuda()
    B = 8
    xT = synth.synthetic_xT(B).cuda()
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        feats = cavp.encode_video(video, normalize=True, pool=False)            # (1,32,512)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        c = m.get_learned_conditioning(feats.repeat(B, 1, 1))
        z, _ = m.sample_log_diff_sampler(c, B, "DDIM", 25, unconditional_guidance_scale=4.5,
                                         unconditional_conditioning=torch.zeros_like(c), x_T=xT)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        mel = m.decode_first_stage(z)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
    print(f"config[4]/GPU [{prec}] 1 video x 8 candidates: CAVP(32x224x224) {1e3 * (t1 - t0):6.2f} ms "
          f"({32 / (t1 - t0):7.0f} frames/s)  sample {1e3 * (t2 - t1):7.1f} ms  decode {1e3 * (t3 - t2):6.1f} ms  "
          f"-> {B / (t3 - t0):6.2f} candidate-clips/s, finite={bool(torch.isfinite(mel).all())}")
    del cavp, m
    torch.cuda.empty_cache()

# ---- the notebook's pre- and post-processing on the device
import numpy as np  # noqa: E402
rng = np.random.default_rng(0)
frames = rng.integers(0, 256, (40, 360, 640, 3), dtype=np.uint8)               # one batch of 40 decoded frames
ft = torch.from_numpy(frames).cuda()
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x = P.frames_to_tensor(ft, (224, 224))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
print(f"frame pre-processing: 40 frames 360x640 -> 224x224 (PIL-exact resize + ToTensor): {1e3 * (t1 - t0):6.2f} ms")
mel = (0.55 + 0.25 * torch.rand(4, 128, 512)).cuda()                           # decode_first_stage(z)[:, 0] of 4 clips
t_nnls, t_gl = 1e9, 1e9
for it in range(5):                      # best of 5: the first calls carry torch's allocator growing its pools
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    S = P.vocoder.mel_to_stft(mel)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    wav = P.vocoder.griffinlim(S)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    t_nnls, t_gl = min(t_nnls, t1 - t0), min(t_gl, t2 - t1)
print(f"mel -> waveform, 4 clips of 8.2 s (inverse_op: 24.4 of the notebook's 30 s on CPU): NNLS {1e3 * t_nnls:6.2f} ms + "
      f"Griffin-Lim x32 {1e3 * t_gl:6.2f} ms = {1e3 * (t_nnls + t_gl):6.2f} ms (best of 5), finite={bool(torch.isfinite(wav).all())}")
