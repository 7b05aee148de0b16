"""Race screen of dependent kernel CHAINS under GPU contention (run two copies at once): single UNet blocks through
df_test_unet_block, the VAE decoder, the cond stage and a whole UNet forward, repeated on identical inputs."""
import hashlib
import os
import sys
from collections import Counter

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd as P  # noqa: E402
from diff_foley_amd import synth  # noqa: E402

label, reps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60
m = P.LatentDiffusion(**P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY))
m.load_state_dict(synth.make_state_dict(synth.state_dict_spec(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY), 0))
m.cuda()
eng = m.engine
g = torch.Generator().manual_seed(5)
rn = lambda *s: torch.randn(*s, generator=g).cuda()


def screen(name, fn):
    hs = Counter()
    for _ in range(reps):
        y = fn()
        torch.cuda.synchronize()
        hs[hashlib.md5(y.float().cpu().numpy().tobytes()).hexdigest()[:8]] += 1
    print(f"{label} {name}: {len(hs)} distinct" + ("" if len(hs) == 1 else f"   <-- NONDETERMINISTIC {sorted(hs.values(), reverse=True)[:6]}"))


semb = F.silu(rn(4, 256))
ctx = rn(4, 32, 128)
for (p, cin, cout, h, w) in [("input_blocks.1.0", 64, 64, 16, 64), ("input_blocks.4.0", 64, 128, 8, 32), ("input_blocks.7.0", 128, 256, 4, 16),
                             ("middle_block.0", 256, 256, 2, 8), ("output_blocks.5.0", 384, 256, 4, 16), ("output_blocks.11.0", 128, 64, 16, 64)]:
    x = rn(4, cin, h, w)
    screen(f"resblock {p} {cin}->{cout} @{h}x{w}", lambda: eng.test_block(p, 0, x, semb=semb, cout=cout))
for (p, c, h, w) in [("input_blocks.1.1", 64, 16, 64), ("input_blocks.4.1", 128, 8, 32), ("input_blocks.7.1", 256, 4, 16), ("middle_block.1", 256, 2, 8)]:
    x = rn(4, c, h, w)
    screen(f"transformer {p} C={c} @{h}x{w}", lambda: eng.test_block(p, 1, x, context=ctx))
x = rn(4, 64, 16, 64)
screen("downsample input_blocks.3.0", lambda: eng.test_block("input_blocks.3.0", 2, x))
x = rn(4, 256, 2, 8)
screen("upsample output_blocks.2.1", lambda: eng.test_block("output_blocks.2.1", 3, x))
feats = rn(4, 32, 64)
screen("cond stage", lambda: m.get_learned_conditioning(feats))
z = rn(2, 4, 16, 64)
screen("vae decode B=2", lambda: m.decode_first_stage(z))
xx, tt = rn(4, 4, 16, 64), torch.tensor([500., 37., 1., 900.]).cuda()
eng.set_context(ctx)
screen("unet forward N=4", lambda: eng.unet_forward(xx, tt))
