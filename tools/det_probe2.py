"""Determinism under GPU sharing: the same tiny sampling run N times while a second process competes for the GPU."""
import sys, torch, hashlib
sys.path.insert(0, sys.argv[3] if len(sys.argv) > 3 else "/root/repo")
import diff_foley_amd as P
from diff_foley_amd import synth
cfg = P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY)
m = P.LatentDiffusion(**cfg)
m.load_state_dict(synth.make_state_dict(synth.state_dict_spec(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY), 0))
m.cuda()
feats = synth.synthetic_cavp(4, 32, 64, seed=1234)[:2].cuda()
xT = synth.synthetic_xT(2).cuda()
outs = []
for it in range(int(sys.argv[1])):
    c = m.get_learned_conditioning(feats)
    z, _ = m.sample_log_diff_sampler(c, 2, "DDIM", 6, unconditional_guidance_scale=4.5, unconditional_conditioning=torch.zeros_like(c), x_T=xT)
    mel = m.decode_first_stage(z)[:, 0]
    outs.append((hashlib.md5(z.cpu().numpy().tobytes()).hexdigest()[:8], hashlib.md5(mel.cpu().numpy().tobytes()).hexdigest()[:8]))
from collections import Counter
print(sys.argv[2], "distinct (z, mel) hashes:", Counter(outs))
