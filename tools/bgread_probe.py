"""Premise test for a background weight prefetcher ("weights of the next op prefetched into the Infinity Cache"):
what does a throttled streaming read on a SECOND stream cost the denoise step, before it buys anything?  The main stream runs N = 8
UNet forwards back to back; the side stream runs diag.hip's streaming-read kernel (16-byte loads, 4 in flight per thread) over a 2 GiB
buffer with 8 ... 128 blocks of 256 threads.  Prints ms per step alone and beside the reader, and the reader's rate while the step runs.
usage: python tools/bgread_probe.py [steps]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd as P  # noqa: E402
from diff_foley_amd import engine as E, synth  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = 8
dev = torch.device("cuda", 0)
sd = synth.make_state_dict(synth.state_dict_spec(), 0)
m = P.LatentDiffusion(**P.stage2_config())
m.load_state_dict(sd)
m.cuda(dev)
m.autotune(True)
feats = synth.synthetic_cavp(B).to(dev)
x = synth.synthetic_xT(B).to(dev)
t = torch.full((B,), 500.0, device=dev)
c = m.get_learned_conditioning(feats)
m.engine.set_context(c)
for _ in range(5):
    m.engine.unet_forward(x, t)
torch.cuda.synchronize()
L = E.lib()
side = torch.cuda.Stream(dev)
buf = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
buf.zero_()
sink = torch.zeros(4, device=dev)


def steps():
    t0 = time.perf_counter()
    for _ in range(STEPS):
        m.engine.unet_forward(x, t)
    torch.cuda.current_stream().synchronize()
    return (time.perf_counter() - t0) / STEPS * 1e3


base = [steps() for _ in range(3)]
print(f"alone: {min(base):.3f} ms/step ({', '.join(f'{b:.3f}' for b in base)})")
for blocks in (4, 8, 16, 32, 64, 128):
    # reader alone: rate
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record()
        L.df_test_peak(2, C.c_void_p(buf.data_ptr()), C.c_void_p(sink.data_ptr()), C.c_size_t(buf.numel()), blocks, C.c_void_p(side.cuda_stream))
        e1.record()
    side.synchronize()
    alone_rate = buf.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e12
    # enough reader launches to cover the main stream's run
    est_ms = min(base) * STEPS * 1.6
    n_launch = max(2, int(est_ms / e0.elapsed_time(e1)) + 2)
    with torch.cuda.stream(side):
        e0.record()
        for _ in range(n_launch):
            L.df_test_peak(2, C.c_void_p(buf.data_ptr()), C.c_void_p(sink.data_ptr()), C.c_size_t(buf.numel()), blocks, C.c_void_p(side.cuda_stream))
        e1.record()
    ms = steps()
    still_running = not e1.query()
    side.synchronize()
    rate = n_launch * buf.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e12
    print(f"reader {blocks:3d} blocks: alone {alone_rate:.2f} TB/s | step {ms:.3f} ms ({(ms / min(base) - 1) * 100:+.1f} %), reader beside it {rate:.2f} TB/s"
          f"{'' if still_running else '  (reader finished before the steps did: rate is an upper bound)'}")
