#!/usr/bin/env python
"""Does a run of ANOTHER plan of the same engine change what a plan computes?  CFG step, non-CFG step (other plan, other context),
CFG step again on the same inputs: the two CFG outputs must be bit-identical; per-op checksums name the first op that is not."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import diff_foley_amd as P
from diff_foley_amd import synth
from helpers import tiny_state_dict

m = P.LatentDiffusion(precision="bf16", **P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY))
m.load_state_dict(tiny_state_dict())
m.cuda()
B = 2
eng = m.engine
x = synth.synthetic_xT(B, seed=5).cuda()
c = m.get_learned_conditioning(synth.synthetic_cavp(B, 32, 64, seed=1234).cuda())
cat = torch.cat([torch.zeros_like(c), c])
t = torch.full((B,), 961.0, device="cuda")


def cfg_step(xx=None):
    eng.set_context(cat)
    eng.debug_checksums(True, 1 << 15)
    o = eng.unet_forward_cfg(x if xx is None else xx, t, 4.5).clone()
    return o, eng.debug_checksums_read()


o1, s1 = cfg_step()
o1b, s1b = cfg_step()
eng.set_context(c)
eng.debug_checksums(False)
n1 = eng.unet_forward(x * 0.5, t * 0.5).clone()
o2, s2 = cfg_step()
x2 = synth.synthetic_xT(B, seed=77).cuda() * 1.7
cfg_step(x2)
o3, s3 = cfg_step()
print("same plan, other latent in between: equal =", torch.equal(o1, o3), " max |diff| %.3e" % float((o1 - o3).abs().max()))
s2 = s3
print("repeat without the other plan in between: equal =", torch.equal(o1, o1b))
print("with the other plan in between: equal =", torch.equal(o1, o2), " max |diff| %.3e" % float((o1 - o2).abs().max()))
if s1 != s2:
    n = min(len(s1), len(s2))
    first = next((i for i in range(n) if s1[i] != s2[i]), n)
    print("first diverging checksum", first, "of", n, ":", eng.debug_checksum_label(first), "| number differing:", sum(1 for i in range(n) if s1[i] != s2[i]))
    for i in range(max(0, first - 3), min(n, first + 3)):
        print("   ", i, eng.debug_checksum_label(i), s1[i] == s2[i])
