"""Probe 2: do two half-batch chains overlap on the GPU when nothing but the GPU can serialise them?
  a) one chain N=B, 25 steps                       (reference)
  b) ONE host thread issues step k of chain 0 (stream 0) then step k of chain 1 (stream 1), no per-step sync
  c) two host threads, free running (no barriers), one stream each
Compare with two PROCESSES sharing the GPU (DF_DIST_SHARE_GPU0=1 torchrun ... bench.py --gpus 2 --batch B/2).
usage: python tools/dual_stream_probe2.py [B]"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd as P  # noqa: E402
from diff_foley_amd import synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sd = synth.make_state_dict(synth.state_dict_spec(), 0)
dev = torch.device("cuda", 0)
STEPS = 25


def mk():
    m = P.LatentDiffusion(**P.stage2_config())
    m.load_state_dict(sd)
    m.cuda(dev)
    m.autotune(True)
    return m


feats = synth.synthetic_cavp(B).to(dev)
x = synth.synthetic_xT(B).to(dev)
t = torch.full((B,), 500.0, device=dev)
m0, m1 = mk(), mk()
c = m0.get_learned_conditioning(feats)
m0.engine.set_context(torch.zeros_like(c))
m1.engine.set_context(c)
models = [m0, m1]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
for i in range(2):
    with torch.cuda.stream(streams[i]):
        for _ in range(3):
            models[i].engine.unet_forward(x, t)
torch.cuda.synchronize()

with torch.cuda.stream(streams[0]):
    t0 = time.perf_counter()
    for _ in range(STEPS):
        m0.engine.unet_forward(x, t)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / STEPS * 1e3
print(f"a) one chain N={B}: {one:.3f} ms/step   (host issue time {t_issue / STEPS * 1e3:.3f} ms/step)")

t0 = time.perf_counter()
for _ in range(STEPS):
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            models[i].engine.unet_forward(x, t)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
alt = (time.perf_counter() - t0) / STEPS * 1e3
print(f"b) two chains, one host thread alternating: {alt:.3f} ms per step pair (host issue {t_issue / STEPS * 1e3:.3f})  x{2 * one / alt:.2f} overlap")


def worker(i):
    torch.cuda.set_device(dev)
    with torch.cuda.stream(streams[i]):
        for _ in range(STEPS):
            models[i].engine.unet_forward(x, t)
        streams[i].synchronize()


th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
t0 = time.perf_counter()
for h in th:
    h.start()
for h in th:
    h.join()
thr = (time.perf_counter() - t0) / STEPS * 1e3
print(f"c) two chains, two free-running host threads: {thr:.3f} ms per step pair  x{2 * one / thr:.2f} overlap")
