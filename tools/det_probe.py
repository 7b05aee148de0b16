import os, sys, torch
sys.path.insert(0, "/root/repo")
import diff_foley_amd as P
from diff_foley_amd import synth
cfg = P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY)
m = P.LatentDiffusion(**cfg)
m.load_state_dict(synth.make_state_dict(synth.state_dict_spec(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY), 0))
m.cuda()
def sample(lo, hi, steps=6):
    feats = synth.synthetic_cavp(4, 32, 64, seed=1234)[lo:hi].cuda()
    xT = synth.synthetic_xT(hi - lo, first_index=lo).cuda()
    c = m.get_learned_conditioning(feats)
    z, _ = m.sample_log_diff_sampler(c, hi - lo, "DDIM", steps, unconditional_guidance_scale=4.5,
                                     unconditional_conditioning=torch.zeros_like(c), x_T=xT)
    return z.cpu()
a = sample(0, 2); b = sample(0, 2); c = sample(2, 4); d = sample(0, 2); e = sample(2, 4)
print("run1 vs run2 (same inputs):", float((a - b).abs().max()), " run2 vs run4:", float((b - d).abs().max()), " shard1 run1 vs run2:", float((c - e).abs().max()))
# single forward determinism
x = torch.randn(4, 4, 16, 64).cuda(); t = torch.tensor([500., 37., 1., 900.]).cuda(); ctx = torch.randn(4, 32, 128).cuda()
m.engine.set_context(ctx)
y = [m.engine.unet_forward(x, t).cpu() for _ in range(4)]
print("forward repeat max|d|:", [float((y[0] - yi).abs().max()) for yi in y[1:]])
z = synth.synthetic_xT(2).cuda()
d = [m.decode_first_stage(z).cpu() for _ in range(4)]
print("decode repeat max|d|:", [float((d[0] - di).abs().max()) for di in d[1:]])
z4 = synth.synthetic_xT(4).cuda()
d4 = m.decode_first_stage(z4).cpu()
d2 = m.decode_first_stage(z).cpu()
print("decode after another batch size:", float((d[0] - d2).abs().max()), " B=4 rows 0:2 vs B=2:", float((d4[:2] - d2).abs().max()))
