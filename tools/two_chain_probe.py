"""Do two HALF-batch UNet chains overlap inside one process when the host issues nothing per kernel?  Two engines (own weights, plans,
workspaces), one captured UNet call each (HIP graph), replayed K times on two streams, against one chain of the full batch.
(Round 2's versions of this question were host-bound -- 9.4 us of host time per launch -- or used two PROCESSES, which time-slice.)
usage: python tools/two_chain_probe.py [B_total]"""
import os
import sys
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


def prep(m, B, first):
    feats = synth.synthetic_cavp(BT)[first:first + B].to(dev)
    x = synth.synthetic_xT(BT)[first:first + B].contiguous().to(dev)
    c = m.get_learned_conditioning(feats)
    m.engine.set_context(torch.cat([torch.zeros_like(c), c]))
    tb = DDIMTables(m.alphas_cumprod, 25)
    steps = np.flip(tb.timesteps)
    m.engine.set_timesteps([float(v) for v in steps], B, 16, 64, True)
    t = torch.full((B,), float(steps[12]), device=dev)
    out = torch.empty_like(x)
    return x, t, out


def capture(m, x, t, out, s):
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            m.engine.unet_forward_cfg(x, t, 4.5, out=out, ts_index=12)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        m.engine.unet_forward_cfg(x, t, 4.5, out=out, ts_index=12)
    torch.cuda.synchronize()
    return g


def timed(fn):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best / K * 1e3


m0, m1 = mk(), mk()
s0, s1 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
# one chain of the whole batch
x, t, out_full = prep(m0, BT, 0)
g = capture(m0, x, t, out_full, s0)


def one():
    with torch.cuda.stream(s0):
        for _ in range(K):
            g.replay()


t_full = timed(one)
print(f"one chain B={BT}: {t_full:.3f} ms per step", flush=True)
ref = out_full.clone()
del g
h = BT // 2
xa, ta, oa = prep(m0, h, 0)
xb, tb_, ob = prep(m1, h, h)
ga, gb = capture(m0, xa, ta, oa, s0), capture(m1, xb, tb_, ob, s1)


def half_alone():
    with torch.cuda.stream(s0):
        for _ in range(K):
            ga.replay()


def two():
    for _ in range(K):
        with torch.cuda.stream(s0):
            ga.replay()
        with torch.cuda.stream(s1):
            gb.replay()


t_half = timed(half_alone)
t_two = timed(two)
same = bool((torch.cat([oa, ob]) == ref).all())
print(f"one chain B={h} alone: {t_half:.3f} ms per step")
print(f"two chains B={h} + B={h} on two streams (graph replays): {t_two:.3f} ms per step pair  -> x{t_full / t_two:.3f} against the single "
      f"B={BT} chain; outputs bit-equal to the B={BT} chain's rows: {same}")
