"""configs[2] (B = 8, 50-step DPM-Solver++ with the double-guidance classifier): classifier gradient on the UNet step's stream
(DF_CLS_OVERLAP=0) against a second stream beside it (=1); latents compared bit for bit.
usage: python tools/cls_overlap_probe.py [B]"""
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
c = m.get_learned_conditioning(feats[:, :32])
uc = torch.zeros_like(c)
out = {}
for rnd in range(2):
    for mode in ("0", "1"):
        os.environ["DF_CLS_OVERLAP"] = mode
        best = 1e9
        for it in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            z, _ = m.sample_log_with_classifier_diff_sampler(c, origin_cond=feats, batch_size=B, sampler_name="DPM_Solver", ddim_steps=50,
                                                             unconditional_guidance_scale=4.5, unconditional_conditioning=uc,
                                                             classifier=cls, classifier_guide_scale=50.0, x_T=xT)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        out.setdefault(mode, z.clone())
        print(f"DF_CLS_OVERLAP={mode}: {best * 1e3:7.1f} ms ({50 / best:6.1f} steps/s)  finite={bool(torch.isfinite(z).all())}  "
              f"== serial: {bool((z == out['0']).all())}", flush=True)
