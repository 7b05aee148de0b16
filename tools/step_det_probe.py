#!/usr/bin/env python
"""Run-to-run determinism of the denoise step: the same (x, t, context) R times through one plan, number of distinct outputs.
usage: det_probe.py [tiny|full] [reps] [precision]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd as P
from diff_foley_amd import synth

which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
if which == "tiny":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from helpers import tiny_state_dict
    m = P.LatentDiffusion(precision=prec, **P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY))
    m.load_state_dict(tiny_state_dict())
else:
    m = P.LatentDiffusion(precision=prec, **P.stage2_config())
    m.load_state_dict(synth.make_state_dict(synth.state_dict_spec(), 0))
m.cuda()
if len(sys.argv) > 4:
    m.autotune(True)
B = 2 if which == "tiny" else 4
eng = m.engine
x = synth.synthetic_xT(B, seed=5).cuda()
c = m.get_learned_conditioning(synth.synthetic_cavp(B, 32, 64 if which == "tiny" else 512, seed=1234).cuda())
eng.set_context(torch.cat([torch.zeros_like(c), c]))
outs = []
for r in range(reps):
    t = torch.full((B,), 961.0 if r % 2 == 0 else 41.0, device="cuda")        # alternate inputs: stale state would show
    o = eng.unet_forward_cfg(x, t, 4.5).clone()
    outs.append(o)
torch.cuda.synchronize()
for par in (0, 1):
    ref = outs[par]
    nd = sum(1 for o in outs[par::2] if not torch.equal(o, ref))
    md = max(float((o - ref).abs().max()) for o in outs[par::2])
    print(f"{which} {prec} t-parity {par}: {nd} of {len(outs[par::2])} runs differ from the first, max |diff| {md:.3e}")
