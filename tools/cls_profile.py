#!/usr/bin/env python
"""Per-op HIP-event profile of the classifier-guidance gradient plan (B = 8, configs[2])."""
import collections, csv, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd as P
from diff_foley_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = P.LatentDiffusion(**P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY))
m.load_state_dict(synth.make_state_dict(synth.state_dict_spec(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY), 0))
m.cuda()
if os.environ.get("CLS_TUNE"): m.autotune(True)
cls = P.AlignmentClassifier(classifier_config=dict(params=dict(synth.CLS_FULL)))
cls.load_state_dict(synth.make_state_dict(synth.classifier_spec(synth.CLS_FULL), 0))
cls.attach(m)
x = synth.synthetic_xT(B).cuda(); t = torch.full((B,), 500.0, device="cuda"); vf = synth.synthetic_cavp(B, 33).cuda()
for _ in range(3):
    g = cls.log_prob_grad(x, t, vf)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    g = cls.log_prob_grad(x, t, vf)
e1.record(); torch.cuda.synchronize()
print(f"classifier_grad B={B}: {e0.elapsed_time(e1) / 20:.3f} ms per call")
m.engine.profile_begin()
for _ in range(5):
    g = cls.log_prob_grad(x, t, vf)
m.engine.profile_end()
m.engine.profile_dump("/tmp/cls_ops.csv")
rows = list(csv.DictReader(open("/tmp/cls_ops.csv")))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    agg[r["tag"]][0] += 1; agg[r["tag"]][1] += float(r["ms"])
tot = sum(v[1] for v in agg.values()) / 5
print(f"instrumented total {tot:.3f} ms per call, {len(rows) // 5} ops")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {k:22s} n={v[0] // 5:3d}  {v[1] / 5 * 1e3:8.1f} us  ({v[1] / v[0] * 1e3:6.1f} us each)")
