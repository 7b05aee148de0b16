#!/usr/bin/env python
"""Which launch of the FULL-SIZE UNet plan is not reproducible?  bf16 (or HUNT_PREC) build, B = 4 (or HUNT_B) CFG forward (N = 2B), the
shipped plan table (DF_TUNED_DEFAULTS=0: the cost-model plans; HUNT_TUNE=1: tuned here); the same forward is repeated with per-op workspace checksums (df_debug_checksums) and every repetition's
sequence is compared with the first one: the first differing op is the launch that produced different bytes from identical inputs.
usage: [DF_LIB_OVERRIDE=...] tools/race_hunt.py <reps> [--prec bf16]"""
import os, sys, json
from collections import Counter
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd as P
from diff_foley_amd import synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
prec = os.environ.get("HUNT_PREC", "bf16")
m = P.LatentDiffusion(precision=prec, **P.stage2_config())
m.load_state_dict(synth.make_state_dict(synth.state_dict_spec(), 0))
m.cuda()
if os.environ.get("HUNT_TUNE"): m.autotune(True)
B = int(os.environ.get("HUNT_B", "4"))
xT = synth.synthetic_xT(B, seed=21).cuda()
c = m.get_learned_conditioning(synth.synthetic_cavp(B, 32, 512, seed=1234).cuda())
uc = torch.zeros_like(c)
eng = m.engine
eng.set_context(torch.cat([uc, c]))
t = torch.full((B,), 961.0, device="cuda")
eng.unet_forward_cfg(xT, t, 4.5)
torch.cuda.synchronize()
ref = None
first_ops = Counter()
nonfinite = 0
for it in range(reps):
    eng.debug_checksums(True, 1 << 12)
    y = eng.unet_forward_cfg(xT, t, 4.5)
    torch.cuda.synchronize()
    seq = eng.debug_checksums_read()
    fin = bool(torch.isfinite(y).all())
    nonfinite += (not fin)
    if ref is None:
        ref = seq
        print("ops per run:", len(seq), "finite:", fin, flush=True)
        continue
    if seq != ref:
        n = min(len(seq), len(ref))
        first = next((i for i in range(n) if seq[i] != ref[i]), n)
        lab = eng.debug_checksum_label(first)
        nd = sum(1 for i in range(n) if seq[i] != ref[i])
        prev = eng.debug_checksum_label(first - 1) if first > 0 else ""
        # an op whose predecessor left the WHOLE workspace identical to the reference run's and which itself did not: that launch
        # produced different bytes from identical inputs (op 0 differing only says the previous run's leftovers differed)
        trans = [i for i in range(1, n) if seq[i] != ref[i] and seq[i - 1] == ref[i - 1]]
        tl = [eng.debug_checksum_label(i) for i in trans[:4]]
        print(f"rep {it}: first differing op {first}: {lab}  differing ops {nd}  finite {fin}  match->differ at {trans[:8]}: {tl}", flush=True)
        for l in tl[:1]:
            first_ops[l] += 1
        if not tl:
            first_ops[lab] += 1
eng.debug_checksums(False)
print(json.dumps({"lib": os.environ.get("DF_LIB_OVERRIDE", "product"), "prec": prec, "B": B, "reps": reps, "diverged": sum(first_ops.values()),
                  "nonfinite_outputs": nonfinite, "first_ops": dict(first_ops)}))
