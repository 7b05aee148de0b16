"""Feature-operand reuse of the classifier gradient (df_classifier_grad_cached, DF_CLS_FEAT_CACHE=0 restores one recomputation per
call): the isolated gradient call and configs[2] (B = 8, 50-step DPM-Solver++ with the double-guidance classifier) on one stream and
with the gradient on the side stream, alternated on one box; latents compared bit for bit.
usage: python tools/cls_feat_cache_probe.py [B]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd as P  # noqa: E402
from diff_foley_amd import synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = P.LatentDiffusion(**P.stage2_config())
m.load_state_dict(synth.make_state_dict(synth.state_dict_spec(), 0))
m.cuda()
cls = P.AlignmentClassifier(classifier_config=dict(params=dict(synth.CLS_FULL)))
cls.load_state_dict(synth.make_state_dict(synth.classifier_spec(synth.CLS_FULL), 0))
cls.attach(m)
feats = synth.synthetic_cavp(B, 33).cuda()
xT = synth.synthetic_xT(B).cuda()
t = torch.full((B,), 500.0, device="cuda")
c = m.get_learned_conditioning(feats[:, :32])
uc = torch.zeros_like(c)

print(f"isolated gradient call, B = {B} (20 calls, best of 3 rounds, alternated)")
ref = None
for rnd in range(3):
    for cache in ("0", "1"):
        os.environ["DF_CLS_FEAT_CACHE"] = cache
        for _ in range(3):
            g = cls.log_prob_grad(xT, t, feats)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g = cls.log_prob_grad(xT, t, feats)
        e1.record()
        torch.cuda.synchronize()
        ref = g.clone() if ref is None else ref
        print(f"  DF_CLS_FEAT_CACHE={cache}: {e0.elapsed_time(e1) / 20:.4f} ms per call   == first: {bool(torch.equal(g, ref))}", flush=True)

print(f"configs[2], B = {B}, DPM-Solver++ 50 steps (best of 3, alternated)")
out = None
for rnd in range(2):
    for overlap in ("0", "1"):
        for cache in ("0", "1"):
            os.environ["DF_CLS_OVERLAP"] = overlap
            os.environ["DF_CLS_FEAT_CACHE"] = cache
            best = 1e9
            for it in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                z, _ = m.sample_log_with_classifier_diff_sampler(c, origin_cond=feats, batch_size=B, sampler_name="DPM_Solver",
                                                                 ddim_steps=50, unconditional_guidance_scale=4.5,
                                                                 unconditional_conditioning=uc, classifier=cls,
                                                                 classifier_guide_scale=50.0, x_T=xT)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            out = z.clone() if out is None else out
            print(f"  DF_CLS_OVERLAP={overlap} DF_CLS_FEAT_CACHE={cache}: {best * 1e3:7.1f} ms ({50 / best:6.1f} steps/s)  "
                  f"finite={bool(torch.isfinite(z).all())}  == first: {bool(torch.equal(z, out))}", flush=True)
