"""Is a denoise step bound by the HOST's launch rate?  For sampler batches B in argv (default 1 2 4 8):
  a) K steps issued with no sync: host issue time per step (wall clock until the last launch call returns) and the
     synchronised time per step;
  b) the same step captured ONCE into a HIP graph (hipStreamBeginCapture through torch.cuda.CUDAGraph: a step is nothing but
     kernel launches on the current stream) and replayed K times: what the GPU needs when the host issues nothing per kernel.
usage: python tools/host_issue_probe.py [B ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd as P  # noqa: E402
from diff_foley_amd import engine as E  # noqa: E402
from diff_foley_amd import synth  # noqa: E402
from diff_foley_amd.schedule import DDIMTables  # noqa: E402

Bs = [int(v) for v in sys.argv[1:]] or [1, 2, 4, 8]
K = int(os.environ.get("K", "50"))
dev = torch.device("cuda", 0)
sd = synth.make_state_dict(synth.state_dict_spec(), 0)
model = P.LatentDiffusion(**P.stage2_config())
model.load_state_dict(sd)
model.cuda(dev)
eng = model.engine
tb = DDIMTables(model.alphas_cumprod, 25)
steps = np.flip(tb.timesteps)

for B in Bs:
    feats = synth.synthetic_cavp(B).to(dev)
    x0 = synth.synthetic_xT(B).to(dev)
    c = model.get_learned_conditioning(feats)
    eng.set_context(torch.cat([torch.zeros_like(c), c]))
    eng.set_timesteps([float(v) for v in steps], B, 16, 64, True)
    t_all = torch.tensor(steps.copy(), dtype=torch.float32, device=dev)[:, None].expand(25, B).contiguous()
    out = torch.empty_like(x0)

    def step(i, x):
        i = i % 25
        idx = 25 - i - 1
        e = eng.unet_forward_cfg(x, t_all[i], 4.5, ts_index=i)
        xn, _ = E.ddim_update(x, e, tb.alphas[idx], tb.alphas_prev[idx], 0.0, tb.sqrt_one_minus_alphas[idx])
        return xn

    def unet_only(i, x):
        return eng.unet_forward_cfg(x, t_all[i % 25], 4.5, out=out, ts_index=i % 25)

    for name, fn in (("step (UNet + DDIM update, python)", step), ("UNet call only", unet_only)):
        x = x0
        for i in range(5):
            r = fn(i, x)
        torch.cuda.synchronize()
        best = None
        for rep in range(3):
            x = x0
            t0 = time.perf_counter()
            for i in range(K):
                r = fn(i, x)
                if fn is step:
                    x = r
            t_issue = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_all_ = time.perf_counter() - t0
            if best is None or t_all_ < best[1]:
                best = (t_issue, t_all_)
        print(f"B={B} {name}: host issue {best[0] / K * 1e3:.3f} ms/step, synchronised {best[1] / K * 1e3:.3f} ms/step", flush=True)

    # graph replay of ONE captured UNet call (timestep row 12; same kernels every step, only the table row differs)
    try:
        s = torch.cuda.Stream(dev)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for i in range(3):
                unet_only(12, x0)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            unet_only(12, x0)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        ref = out.clone()
        unet_only(12, x0)
        torch.cuda.synchronize()
        same = bool((ref == out).all())
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            for i in range(K):
                g.replay()
            t_issue = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_all_ = time.perf_counter() - t0
            if best is None or t_all_ < best[1]:
                best = (t_issue, t_all_)
        print(f"B={B} graph replay of one UNet call: host issue {best[0] / K * 1e3:.3f} ms/step, synchronised "
              f"{best[1] / K * 1e3:.3f} ms/step, bit-equal to the launches: {same}", flush=True)
        del g
    except Exception as ex:  # noqa: BLE001
        print(f"B={B} graph capture failed: {type(ex).__name__}: {ex}", flush=True)
