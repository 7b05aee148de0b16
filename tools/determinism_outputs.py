#!/usr/bin/env python
"""Output-level reproducibility of the plans the shipped table covers besides the B = 4 CFG step (tools/race_hunt.py checks that one
op by op): the same call repeated REPS times must return the same bytes.  UNet CFG / plain forward at several batches, VAE decode,
classifier probability + gradient, CAVP encoder.
usage: [HUNT_PREC=bf16|fp16] tools/determinism_outputs.py [reps]      (exit code 1 on any difference)"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd as P
from diff_foley_amd import synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prec = os.environ.get("HUNT_PREC", "bf16")
m = P.LatentDiffusion(precision=prec, **P.stage2_config())
m.load_state_dict(synth.make_state_dict(synth.state_dict_spec(), 0))
m.cuda()
eng = m.engine
bad = {}


def check(name, fn):
    ref, nd = None, 0
    for _ in range(reps):
        out = fn()
        torch.cuda.synchronize()
        out = [o.clone() for o in (out if isinstance(out, (tuple, list)) else [out])]
        if ref is None:
            ref = out
        elif any(not torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(out, ref)):
            nd += 1
    fin = all(bool(torch.isfinite(o).all()) for o in ref)
    print(f"{name}: {'DIFFERS %d/%d' % (nd, reps - 1) if nd else 'ok'}{'' if fin else '  NON-FINITE'}", flush=True)
    if nd or not fin:
        bad[name] = nd


for B in [int(b) for b in os.environ.get("HUNT_BATCHES", "1,2,3,5,6,8,16").split(",")]:
    xT = synth.synthetic_xT(B, seed=21).cuda()
    c = m.get_learned_conditioning(synth.synthetic_cavp(B, 32, 512, seed=1234).cuda())
    t = torch.full((B,), 961.0, device="cuda")
    eng.set_context(torch.cat([torch.zeros_like(c), c]))
    check(f"unet cfg B={B}", lambda: eng.unet_forward_cfg(xT, t, 4.5))
    eng.set_context(c)
    check(f"unet plain N={B}", lambda: eng.unet_forward(xT, t))
for B in (1, 4):
    z = synth.synthetic_xT(B, seed=5).cuda()
    check(f"vae decode B={B}", lambda: eng.vae_decode(z))
cls = P.AlignmentClassifier(classifier_config=dict(params=dict(synth.CLS_FULL)))
cls.load_state_dict(synth.make_state_dict(synth.classifier_spec(synth.CLS_FULL), 0))
cls.attach(m)
for B in (1, 4, 8):
    x = synth.synthetic_xT(B).cuda(); t = torch.full((B,), 500.0, device="cuda"); vf = synth.synthetic_cavp(B, 33).cuda()
    check(f"classifier grad B={B}", lambda: cls.log_prob_grad(x, t, vf))
print(json.dumps({"prec": prec, "reps": reps, "differing": bad}))
sys.exit(1 if bad else 0)
