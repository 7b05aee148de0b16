#!/usr/bin/env python
"""Localise run-to-run differences of the engine: the same tiny sampling run (cond -> DDIM x CFG -> decode) is repeated
with the per-op workspace checksums on (df_debug_checksums), alone or with a SECOND process competing for the GPU, and
every repetition's checksum sequence is compared with the first one.  Reports, per diverging repetition, the FIRST op whose
checksum differs (= the launch that was not reproducible; everything before it saw identical bytes).

usage: tools/chk_probe.py <reps> [--partner] [--steps S] [--out FILE]
  --partner   spawn a second process running the same loop at the same time (the shared-GPU situation of
              tests/test_multi_rank_gpu.py); both processes report."""
import argparse
import hashlib
import json
import os
import subprocess
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(reps, steps, label, mode):
    import torch
    import diff_foley_amd as P
    from diff_foley_amd import synth
    cfg = P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY)
    m = P.LatentDiffusion(**cfg)
    m.load_state_dict(synth.make_state_dict(synth.state_dict_spec(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY), 0))
    m.cuda()
    feats = synth.synthetic_cavp(4, 32, 64, seed=1234)[:2].cuda()
    xT = synth.synthetic_xT(2).cuda()
    eng = m.engine

    def once():
        c = m.get_learned_conditioning(feats)
        z, _ = m.sample_log_diff_sampler(c, 2, "DDIM", steps, unconditional_guidance_scale=4.5,
                                         unconditional_conditioning=torch.zeros_like(c), x_T=xT)
        mel = m.decode_first_stage(z)[:, 0]
        return hashlib.md5(mel.cpu().numpy().tobytes()).hexdigest()[:10]

    once()                       # builds every plan (first-run effects are reported separately below)
    res = dict(label=label, mode=mode, reps=reps, diverged=[], first_ops=Counter())
    ref = None
    hashes = Counter()
    for it in range(reps):
        if mode == "chk":
            eng.debug_checksums(True, 1 << 15)
        h = once()
        hashes[h] += 1
        if mode != "chk":
            continue
        seq = eng.debug_checksums_read()
        if ref is None:
            ref = seq
            res["ops_per_run"] = len(seq)
            continue
        if seq != ref:
            n = min(len(seq), len(ref))
            first = next((i for i in range(n) if seq[i] != ref[i]), n)
            lab = eng.debug_checksum_label(first) if first < len(seq) else "length"
            ndiff = sum(1 for i in range(n) if seq[i] != ref[i])
            res["diverged"].append(dict(rep=it, first_index=first, first_op=lab, n_differing=ndiff))
            res["first_ops"][lab.split(":", 1)[-1]] += 1
    if mode == "chk":
        eng.debug_checksums(False)
    res["mel_hashes"] = dict(hashes)
    res["first_ops"] = dict(res["first_ops"])
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("reps", type=int)
    ap.add_argument("--partner", action="store_true")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--mode", default="chk", choices=["chk", "hash"])
    ap.add_argument("--label", default="main")
    ap.add_argument("--out", default="")
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    child = None
    if a.partner and not a.child:
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__), str(a.reps), "--steps", str(a.steps), "--mode", a.mode,
                                  "--label", "partner", "--child"], stdout=subprocess.PIPE, text=True)
    res = [run(a.reps, a.steps, a.label, a.mode)]
    if child is not None:
        out, _ = child.communicate(timeout=1800)
        for line in out.splitlines():
            if line.startswith("{"):
                res.append(json.loads(line))
    for r in res:
        print(json.dumps(r))
    if a.out and not a.child:
        with open(a.out, "w") as f:
            for r in res:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
