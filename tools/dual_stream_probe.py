"""Probe: is the UNet step latency-bound enough that two half-batch chains on two HIP streams beat one full-batch chain?
Two engine contexts (own weight copies: pessimistic for L2/MALL sharing) driven by two host threads, one stream each;
per step both halves are joined (as the CFG combine would).  usage: python tools/dual_stream_probe.py [B]"""
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


def mk():
    m = P.LatentDiffusion(**P.stage2_config())
    m.load_state_dict(sd)
    m.cuda(dev)
    m.autotune(True)
    return m


feats = synth.synthetic_cavp(B).to(dev)
x = synth.synthetic_xT(B).to(dev)
t = torch.full((B,), 500.0, device=dev)

# ---- baseline: one chain, N = 2B
m0 = mk()
c = m0.get_learned_conditioning(feats)
uc = torch.zeros_like(c)
m0.engine.set_context(torch.cat([uc, c]))
x2, t2 = torch.cat([x, x]), torch.cat([t, t])
for _ in range(3):
    m0.engine.unet_forward(x2, t2)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(25):
    m0.engine.unet_forward(x2, t2)
torch.cuda.synchronize()
base = (time.perf_counter() - t0) / 25 * 1e3
print(f"one chain  N={2*B}: {base:.3f} ms/step")

# ---- one chain at N = B (how much does halving the batch buy per launch?)
m0.engine.set_context(c)
for _ in range(3):
    m0.engine.unet_forward(x, t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(25):
    m0.engine.unet_forward(x, t)
torch.cuda.synchronize()
half = (time.perf_counter() - t0) / 25 * 1e3
print(f"one chain  N={B}: {half:.3f} ms/step")

# ---- two chains of N = B on two streams, two host threads
m1 = mk()
m0.engine.set_context(uc)
m1.engine.set_context(c)
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
models = [m0, m1]
for i in range(2):
    with torch.cuda.stream(streams[i]):
        for _ in range(3):
            models[i].engine.unet_forward(x, t)
torch.cuda.synchronize()
bar = threading.Barrier(3)


def worker(i, steps):
    torch.cuda.set_device(dev)
    with torch.cuda.stream(streams[i]):
        for _ in range(steps):
            bar.wait()
            models[i].engine.unet_forward(x, t)
            streams[i].synchronize()
            bar.wait()


steps = 25
th = [threading.Thread(target=worker, args=(i, steps)) for i in range(2)]
for h in th:
    h.start()
t0 = time.perf_counter()
for _ in range(steps):
    bar.wait()
    bar.wait()
dual = (time.perf_counter() - t0) / steps * 1e3
for h in th:
    h.join()
print(f"two chains N={B}+{B} on two streams: {dual:.3f} ms/step  (x{base/dual:.2f} vs one chain)")
