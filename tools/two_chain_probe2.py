"""Two HALF-batch UNet chains, plain stream launches (no graphs), one free-running host thread per chain (ctypes releases the GIL
inside the engine call), against one chain of the full batch.  Companion of tools/two_chain_probe.py (graph replays).
usage: python tools/two_chain_probe2.py [B_total]"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd as P  # noqa: E402
from diff_foley_amd import synth  # noqa: E402
from diff_foley_amd.schedule import DDIMTables  # noqa: E402

BT = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = 50
dev = torch.device("cuda", 0)
sd = synth.make_state_dict(synth.state_dict_spec(), 0)


def mk():
    m = P.LatentDiffusion(**P.stage2_config())
    m.load_state_dict(sd)
    m.cuda(dev)
    return m


def prep(m, B):
    feats = synth.synthetic_cavp(B).to(dev)
    x = synth.synthetic_xT(B).to(dev)
    c = m.get_learned_conditioning(feats)
    m.engine.set_context(torch.cat([torch.zeros_like(c), c]))
    tb = DDIMTables(m.alphas_cumprod, 25)
    steps = np.flip(tb.timesteps)
    m.engine.set_timesteps([float(v) for v in steps], B, 16, 64, True)
    t = torch.full((B,), float(steps[12]), device=dev)
    return x, t, torch.empty_like(x)


def chain(m, x, t, out, s, n):
    with torch.cuda.stream(s):
        for _ in range(n):
            m.engine.unet_forward_cfg(x, t, 4.5, out=out, ts_index=12)


def timed(jobs):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        ths = [threading.Thread(target=chain, args=j) for j in jobs]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best / K * 1e3


m0, m1 = mk(), mk()
s0, s1 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
a = prep(m0, BT)
chain(m0, *a, s0, 3)
t_full = timed([(m0, *a, s0, K)])
h = BT // 2
a0, a1 = prep(m0, h), prep(m1, h)
chain(m0, *a0, s0, 3)
chain(m1, *a1, s1, 3)
t_half = timed([(m0, *a0, s0, K)])
t_two = timed([(m0, *a0, s0, K), (m1, *a1, s1, K)])
print(f"B={BT}: one chain {t_full:.3f} ms/step; one B={h} chain {t_half:.3f}; two B={h} chains, two streams, two host threads "
      f"{t_two:.3f} ms per step pair -> x{t_full / t_two:.3f} against the single chain")
