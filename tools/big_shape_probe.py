"""Probe: the full UNet (CFG plan) and VAE decoder on shapes far beyond the BASELINE's -- long latents x large batches -- product only:
finite results, batch rows consistent with a B = 1 run of the same row, or an error that says what is too large.
usage: python tools/big_shape_probe.py [B,W ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd as P  # noqa: E402
from diff_foley_amd import synth  # noqa: E402

m = P.LatentDiffusion(**P.stage2_config())
m.load_state_dict(synth.make_state_dict(synth.state_dict_spec(), 0))
m.cuda()
cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(4, 256), (1, 1024), (8, 512), (16, 1024)]
for B, W in cases:
    g = torch.Generator().manual_seed(B * 10000 + W)
    x = torch.randn(B, 4, 16, W, generator=g).cuda()
    c = (torch.randn(B, 32, 768, generator=g) * 0.05).cuda()
    t = torch.full((B,), 481.0).cuda()
    for name, fn in (("unet cfg", lambda xx, cc, tt: (m.engine.set_context(torch.cat([torch.zeros_like(cc), cc])),
                                                       m.engine.unet_forward_cfg(xx, tt, 4.5))[1]),
                     ("vae", lambda xx, cc, tt: m.decode_first_stage(xx))):
        try:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            y = fn(x, c, t)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            y1 = fn(x[:1].contiguous(), c[:1].contiguous(), t[:1])
            torch.cuda.synchronize()
            rel = float((y[:1] - y1).norm() / y1.norm())
            print(f"B={B} W={W} {name}: finite={bool(torch.isfinite(y).all())} row 0 vs its B = 1 run rel-L2 {rel:.2e}  ({dt * 1e3:.0f} ms first call)",
                  flush=True)
        except RuntimeError as ex:
            print(f"B={B} W={W} {name}: RuntimeError: {str(ex)[:300]}", flush=True)
        m._ctx_owner = None
