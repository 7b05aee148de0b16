#!/usr/bin/env python
"""bench.py -- denoise steps/s of the Stage-2 sampling path on MI355X (BASELINE.json metric).

One "step" = one DDIM denoise step of one batch: CFG UNet evaluation on N = 2B latents (batch duplication, UNet
forward, guidance combine) + the DDIM update, exactly the body of ddim.py:206-227.  Workload = BASELINE.json
configs[1]: B = 4 per GPU, 8 s audio latent (4x16x64), 32 CAVP context frames, guidance 4.5, bf16 MFMA operands,
procedurally generated weights of the full 860 M-parameter UNet (no checkpoint is reachable offline).

    python bench.py --gpus 1 --steps 25 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import diff_foley_amd as P  # noqa: E402
from diff_foley_amd import engine as E, parallel, synth  # noqa: E402
from diff_foley_amd.schedule import DDIMTables  # noqa: E402

GFLOP_PER_SAMPLE = 177.86          # one UNet forward (SURVEY.md section 6 / BASELINE.md section 3)
GEMM_GFLOP_PER_SAMPLE = 169.83     # conv 110.99 + Linear 58.85 (the MFMA implicit-GEMM kernel family)
PEAK_BF16_TFLOPS = 2500.0          # dense MFMA bf16 (MI355X_MICROARCH.md)


def cpu_baseline(sd, nsteps):
    """The oracle (CPU restatement of the reference sampler, fp32) timed on this box's host cores:
    BASELINE.json configs[0]: B=1, CFG 4.5 (UNet batch 2), DDIM.  Bounded sample: `nsteps` steps.
    32 threads: measured fastest on the 128-core GPU box (8: 0.5, 16: 0.78, 32: 0.80, 64: 0.46, 128: 0.20 steps/s)."""
    from oracle import unet as ou, vae as ov, schedule as osch
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    usd = ou.sub_state_dict(sd, "model.diffusion_model.")
    csd = ou.sub_state_dict(sd, "cond_stage_model.")
    x = synth.synthetic_xT(1)
    c = ov.cond_stage(csd, synth.synthetic_cavp(1))
    uc = torch.zeros_like(c)
    sch = osch.ddim_schedule(osch.ddpm_schedule()["alphas_cumprod"], 25)
    steps = np.flip(sch["timesteps"])

    def one(i, x):
        idx = 25 - i - 1
        ts = torch.full((2,), int(steps[i]), dtype=torch.long)
        e_u, e_c = ou.unet_forward(usd, synth.UNET_FULL, torch.cat([x, x]), ts, torch.cat([uc, c])).chunk(2)
        e = e_u + 4.5 * (e_c - e_u)
        a_t, a_p = float(sch["alphas"][idx]), float(sch["alphas_prev"][idx])
        p0 = (x - float(sch["sqrt_one_minus_alphas"][idx]) * e) / a_t ** 0.5
        return a_p ** 0.5 * p0 + (1 - a_p) ** 0.5 * e
    x = one(0, x)                      # warm-up (thread pools, allocator)
    t0 = time.perf_counter()
    for i in range(1, 1 + nsteps):
        x = one(i, x)
    dt = time.perf_counter() - t0
    return dict(value=nsteps / dt, unit="denoise_steps/s at B=1 (UNet batch 2)", cores=torch.get_num_threads(),
                kind="port", sample=f"{nsteps} DDIM steps of BASELINE config[0] (B=1, CFG 4.5, fp32, oracle/unet.py), "
                                    f"{dt:.1f} s of CPU work")


def pmc_traffic():
    """HBM-side bytes per GEMM launch (FETCH_SIZE x2 per the gfx950 calibration + WRITE_SIZE, separate --pmc passes of
    this same command; tools/pmc_traffic.sh).  rocprofv3 cannot run inside the timed process, so the number is read
    from the committed profile of the current round; null when absent."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)["gemm_bytes_per_launch"]
    except Exception:
        return None


def measured_peak():
    """MFMA issue peak measured on an MI355X by tools/peaks.py (profiles/peaks.json), next to the nominal one."""
    try:
        with open(os.path.join(ROOT, "profiles", "peaks.json")) as f:
            return json.load(f)["mfma_bf16_tflops"]
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="samples per GPU (UNet batch is 2x this)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=10)
    ap.add_argument("--no-autotune", action="store_true")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16"],
                    help="MFMA operand type (bf16 = BASELINE config 2, the default; fp16 = libdfengine_f16.so)")
    ap.add_argument("--dump-ops", default="", help="write a per-op CSV of the instrumented pass")
    a = ap.parse_args()

    rank, world, local = parallel.init_process_group()
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local)
    B = a.batch
    G = B * world                                           # weak scaling: fixed per-GPU batch

    # ---- weights: generated on rank 0, ONE flat RCCL broadcast over xGMI, re-packed to bf16 on every rank
    spec = synth.state_dict_spec()
    sd = synth.make_state_dict(spec, 0) if rank == 0 else None
    t0 = time.perf_counter()
    if world > 1:
        sd_dev = parallel.broadcast_state_dict(sd, spec, dev, src=0)
        torch.cuda.synchronize()
    else:
        sd_dev = sd
    t_bcast = time.perf_counter() - t0
    model = P.LatentDiffusion(precision=a.precision, **P.stage2_config())
    model.load_state_dict(sd_dev)
    model.cuda(dev)
    if not a.no_autotune:
        model.autotune(True)
    del sd_dev

    # ---- this rank's shard of the global batch (seeded by GLOBAL sample index)
    lo, hi = parallel.shard_range(G, rank, world)
    feats = synth.synthetic_cavp(G)[lo:hi].to(dev)
    x = synth.synthetic_xT(hi - lo, first_index=lo).to(dev)
    c = model.get_learned_conditioning(feats)
    uc = torch.zeros_like(c)
    eng = model.engine
    eng.set_context(torch.cat([uc, c]))
    tb = DDIMTables(model.alphas_cumprod, 25)
    steps = np.flip(tb.timesteps)
    t_all = torch.tensor(steps.copy(), dtype=torch.float32, device=dev)[:, None].expand(25, B).contiguous()

    def step(i, x):
        i = i % 25
        idx = 25 - i - 1
        e = eng.unet_forward_cfg(x, t_all[i], 4.5)
        xn, _ = E.ddim_update(x, e, tb.alphas[idx], tb.alphas_prev[idx], 0.0, tb.sqrt_one_minus_alphas[idx])
        return xn

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        x = step(i, x)
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        x = step(a.warmup + i, x)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(x).all() or os.environ.get("DF_GEMM_DBG")

    # ---- per-kernel-family time, HIP events on the launch stream, same K steps (instrumented pass)
    eng.profile_begin()
    for i in range(a.steps):
        x = step(a.warmup + i, x)
    prof = eng.profile_end()
    sub = None
    if rank == 0:
        import csv
        import tempfile
        dump = a.dump_ops or os.path.join(tempfile.gettempdir(), f"df_ops_{os.getpid()}.csv")
        eng.profile_dump(dump)
        # the dump holds every launch of the K instrumented steps; fold it to ONE step (mean ms per op position)
        rows = list(csv.DictReader(open(dump)))
        if rows and len(rows) % a.steps == 0:
            n = len(rows) // a.steps
            for i in range(n):
                rows[i]["ms"] = "%.5f" % (sum(float(rows[i + k * n]["ms"]) for k in range(a.steps)) / a.steps)
            rows = rows[:n]
            if a.dump_ops:
                with open(dump, "w", newline="") as f:
                    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
                    w.writeheader()
                    w.writerows(rows)
            # the two family metrics the north star names: SpatialTransformer MFMA rate over every op of
            # the 16 transformers (GEMMs, attention, LayerNorm) and ResBlock-conv HBM rate over the 44 conv3x3 launches
            st_ms = sum(float(r["ms"]) for r in rows if r["tag"].startswith(("st.", "attn.")) or r["tag"] == "layernorm")
            rc_ms = sum(float(r["ms"]) for r in rows if r["tag"] in ("res.conv1", "res.conv2"))
            sub = {"st_ms": st_ms, "rc_ms": rc_ms}
        if not a.dump_ops:
            os.remove(dump)
    stats = eng.plan_stats()

    if rank == 0:
        ms_step = dt / a.steps * 1e3
        per_gpu = a.steps / dt
        N = 2 * B
        gemm_ms = prof["gemm"]["ms"] / a.steps
        gemm_launches = prof["gemm"]["launches"] // a.steps
        gemm_tflops = GEMM_GFLOP_PER_SAMPLE * N / gemm_ms            # GFLOP / ms = TFLOP/s
        out = {
            "metric": "UNet denoise steps/sec (8s audio latent, 25-step DDIM, CFG 4.5), aggregate over GPUs",
            "value": round(per_gpu * world, 3),
            "unit": "denoise_steps/s",
            "per_gpu": round(per_gpu, 3),
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if a.precision == "bf16" else "f16",
            "data": "synthetic (procedural weights of the full 859.5M-param UNet, unit-norm CAVP-like features, seeded x_T)",
            "config": {"workload": "BASELINE.json configs[1]: single MI355X, batch=4 (UNet batch 8), 25-step DDIM, "
                                   + ("bf16" if a.precision == "bf16" else "fp16") + " UNet, 32 CAVP context frames, latent 4x16x64, guidance 4.5",
                       "batch_per_gpu": B, "global_batch": G, "parallelism": f"batch-shard x{world}, no step-loop collectives",
                       "weight_bcast_s": round(t_bcast, 3) if world > 1 else None},
            "step_tflops_algorithmic": round(GFLOP_PER_SAMPLE * N / ms_step, 2),
            "roofline": {"bound": "mfma", "kernel": "gemm_bf16_kernel (implicit-GEMM conv3x3/1x1/linear, all tile shapes)",
                         "achieved": round(gemm_tflops, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(gemm_tflops / PEAK_BF16_TFLOPS, 4), "traffic": pmc_traffic(),
                         "peak_measured": measured_peak(),
                         "launches_per_step": int(gemm_launches), "avg_launch_us": round(gemm_ms * 1e3 / max(1, gemm_launches), 2),
                         "algorithmic_gflop_per_step": round(GEMM_GFLOP_PER_SAMPLE * N, 1)},
            "north_star_families": None if not sub else {
                "spatial_transformer": {"algorithmic_gflop_per_step": round(73.22 * N, 1), "ms_per_step": round(sub["st_ms"], 4),
                                        "tflops": round(73.22 * N / sub["st_ms"], 1),
                                        "frac_of_mfma_peak": round(73.22 * N / sub["st_ms"] / PEAK_BF16_TFLOPS, 4)},
                "resblock_conv3x3": {"algorithmic_gb_per_step": round(1.0235 + 0.0403 * N, 4), "ms_per_step": round(sub["rc_ms"], 4),
                                     "gb_per_s": round((1.0235 + 0.0403 * N) / sub["rc_ms"] * 1e3, 1),
                                     "frac_of_hbm_peak": round((1.0235 + 0.0403 * N) / sub["rc_ms"] * 1e3 / 8000.0, 4),
                                     "tflops": round(81.62 * N / sub["rc_ms"], 1)}},
            "kernel_ms_per_step": {k: round(v["ms"] / a.steps, 4) for k, v in prof.items()},
            "kernel_launches_per_step": {k: int(v["launches"] // a.steps) for k, v in prof.items()},
            "plan": stats,
        }
        if not a.no_cpu_baseline and world == 1:       # reported at N=1 only (rank 0's host cores)
            out["cpu_baseline"] = cpu_baseline(sd, a.cpu_steps)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
