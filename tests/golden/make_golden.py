#!/usr/bin/env python
"""Generate the golden vectors that pin the oracle (and, through it, the HIP path).

Runs ONLY in the build container: it imports the reference implementation from
/root/reference (via oracle/ref_import.py stubs) on CPU, feeds it procedurally
generated weights (diff_foley_amd/synth.py) and seeded inputs, and stores the
reference's *outputs* as small .npz fixtures next to this script.  Inputs are
regenerated from seeds by the tests; no reference source is copied.

    python tests/golden/make_golden.py --tiny     # seconds
    python tests/golden/make_golden.py --full     # ~10 min (860 M-param reference on CPU)
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import diff_foley_amd  # noqa: E402,F401
from diff_foley_amd import synth  # noqa: E402
from oracle import ref_import  # noqa: E402


def rnd(shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def save(name, **arrs):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def check_spec(model, spec):
    rsd = model.state_dict()
    for k, s in spec.items():
        assert tuple(rsd[k].shape) == tuple(s), (k, tuple(rsd[k].shape), s)
    ref_keys = [k for k in rsd if k.startswith(("model.diffusion_model", "first_stage_model.decoder",
                                                "first_stage_model.post_quant", "cond_stage_model"))]
    assert set(ref_keys) == set(spec.keys()), set(ref_keys) ^ set(spec.keys())


def g1_schedules(model, ns):
    """G1: DDPM buffers, DDIM tables S in {25,50}, DPM-Solver (t, lambda, alpha, sigma) S in {25,50}."""
    names = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
             "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
             "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
             "posterior_mean_coef1", "posterior_mean_coef2"]
    out = {n: getattr(model, n) for n in names}
    for S in (25, 50):
        s = ns.DDIMSampler(model)
        s.make_schedule(S, ddim_eta=0.0, verbose=False)
        out[f"ddim{S}_timesteps"] = s.ddim_timesteps
        out[f"ddim{S}_alphas"] = np.asarray(s.ddim_alphas, dtype=np.float64)
        out[f"ddim{S}_alphas_prev"] = np.asarray(s.ddim_alphas_prev, dtype=np.float64)
        out[f"ddim{S}_sqrt_one_minus_alphas"] = np.asarray(s.ddim_sqrt_one_minus_alphas, dtype=np.float64)
        out[f"ddim{S}_sigmas"] = np.asarray(s.ddim_sigmas, dtype=np.float64)
        nsch = ns.dpm.NoiseScheduleVP("discrete", alphas_cumprod=model.alphas_cumprod)
        t = torch.linspace(1.0, 1.0 / 1000, S + 1)
        out[f"dpm{S}_t"] = t
        out[f"dpm{S}_lambda"] = nsch.marginal_lambda(t)
        out[f"dpm{S}_alpha"] = nsch.marginal_alpha(t)
        out[f"dpm{S}_sigma"] = nsch.marginal_std(t)
    tq = torch.tensor([0.0005, 0.001, 0.00137, 0.5, 0.73219, 0.999, 1.0, 1.2])
    out["interp_t"] = tq
    out["interp_log_alpha"] = nsch.marginal_log_mean_coeff(tq)
    # G2: timestep embedding, int and fractional t
    te_t = torch.tensor([0.0, 1.0, 41.0, 961.0, 999.0, 500.25, 37.7, 0.999])
    out["temb_t"] = te_t
    out["temb_320"] = ns.util.timestep_embedding(te_t, 320)
    out["temb_64"] = ns.util.timestep_embedding(te_t, 64)
    save("g1_schedules.npz", **out)


def tiny(seed=0):
    spec = synth.state_dict_spec(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY)
    sd = synth.make_state_dict(spec, seed)
    cfg = ref_import.load_ldm_config(unet=synth.UNET_TINY, vae=synth.VAE_TINY, cond=synth.COND_TINY)
    model, ns = ref_import.build_reference_ldm(cfg, sd)
    check_spec(model, spec)
    g1_schedules(model, ns)

    # ---- G3: per-op tensors, tiny config, latent 8x16 (hooks on the reference modules)
    unet = model.model.diffusion_model
    cap = {}

    def hook(name):
        def f(mod, inp, out):
            cap[name + "__in"] = inp[0].detach().clone()
            if len(inp) > 1 and torch.is_tensor(inp[1]):
                cap[name + "__in1"] = inp[1].detach().clone()
            cap[name + "__out"] = out.detach().clone()
        return f
    watch = {
        "input_blocks.1.0": unet.input_blocks[1][0],       # ResBlock C->C  @ full res
        "input_blocks.1.1": unet.input_blocks[1][1],       # SpatialTransformer
        "input_blocks.3.0": unet.input_blocks[3][0],       # Downsample
        "input_blocks.4.0": unet.input_blocks[4][0],       # ResBlock C->2C (skip conv)
        "middle_block.1": unet.middle_block[1],            # ST at lowest res
        "output_blocks.2.1": unet.output_blocks[2][1],     # Upsample
        "output_blocks.5.0": unet.output_blocks[5][0],     # ResBlock on concat input
        "output_blocks.5.1": unet.output_blocks[5][1],     # ST
    }
    hs = [m.register_forward_hook(hook(n)) for n, m in watch.items()]
    x = rnd((2, 4, 8, 16), 100)
    t = torch.tensor([500, 37])
    c = rnd((2, 32, 128), 101)
    with torch.no_grad():
        y = model.apply_model(x, t, c)
    for h in hs:
        h.remove()
    save("g3_tiny_ops.npz", x=x, t=t, c=c, y=y, **cap)

    # tiny UNet at the real latent size, long and float t
    x = rnd((2, 4, 16, 64), 102)
    tf = torch.tensor([500.25, 37.7])
    with torch.no_grad():
        y_int = model.apply_model(x, t, c)
        y_flt = model.apply_model(x, tf, c)
        zdec = model.decode_first_stage(rnd((2, 4, 16, 64), 103))
        cond = model.get_learned_conditioning(rnd((2, 32, 64), 104))
    save("g3_tiny_unet.npz", y_int=y_int, y_flt=y_flt, decode=zdec, cond=cond)

    # ---- tiny sampler trajectories (all three samplers + ancestral), CFG 4.5
    B = 2
    xT = synth.synthetic_xT(B, seed=21)
    feats = synth.synthetic_cavp(B, 32, 64, seed=1234)
    with torch.no_grad():
        c = model.get_learned_conditioning(feats)
        uc = torch.zeros_like(c)
        out = {}
        for name, S in (("DDIM", 25), ("DDIM", 50), ("DPM_Solver", 25), ("DPM_Solver", 10), ("PLMS", 25)):
            z, inter = model.sample_log_diff_sampler(c, B, name, S, unconditional_guidance_scale=4.5,
                                                     unconditional_conditioning=uc, x_T=xT.clone())
            out[f"{name}_{S}_z"] = z
            if inter is not None:
                out[f"{name}_{S}_pred_x0_last"] = inter["pred_x0"][-1]
                out[f"{name}_{S}_n_inter"] = len(inter["x_inter"])
        z, _ = model.sample_log_diff_sampler(c, B, "DDIM", 25, x_T=xT.clone())     # no CFG
        out["DDIM_25_nocfg_z"] = z
        out["DDIM_25_mel"] = model.decode_first_stage(out["DDIM_25_z"])
        # ancestral sampler, 6 steps, noise from a seeded CPU generator (torch.manual_seed)
        torch.manual_seed(77)
        z, inter = model.sample(c, batch_size=B, return_intermediates=True, x_T=xT.clone(), timesteps=6,
                                shape=(B, 4, 16, 64), verbose=False)
        out["ancestral_6_z"] = z
    save("g5_tiny_samplers.npz", **out)

    # ---- G6: classifier prob + grad (tiny), and double-guidance DDIM / DPM trajectories
    cspec = synth.classifier_spec(synth.CLS_TINY)
    csd = synth.make_state_dict(cspec, seed)
    cls_cfg = {k: v for k, v in synth.CLS_TINY.items()}
    cls = ref_import.build_reference_classifier(cls_cfg, csd)
    rsd = cls.state_dict()
    assert set("model." + k for k in rsd) == set(cspec), set("model." + k for k in rsd) ^ set(cspec)

    class Wrap:
        def __call__(self, x, t, video_feat):
            return cls(x, context=video_feat, timesteps=t)
    wrap = Wrap()
    x = rnd((2, 4, 16, 64), 105)
    vf = synth.synthetic_cavp(B, 33, 64, seed=4321)
    tt = torch.tensor([500, 37])
    xin = x.clone().requires_grad_(True)
    p = wrap(xin, t=tt, video_feat=vf)
    g = torch.autograd.grad(torch.log(p).sum(), xin)[0]
    out = {"cls_p": p, "cls_grad": g}
    with torch.no_grad():
        for name, S in (("DDIM", 10), ("DPM_Solver", 10)):
            z, _ = model.sample_log_with_classifier_diff_sampler(
                c, origin_cond=vf, batch_size=B, sampler_name=name, ddim_steps=S,
                unconditional_guidance_scale=4.5, unconditional_conditioning=uc,
                classifier=wrap, classifier_guide_scale=50.0, x_T=xT.clone())
            out[f"{name}_{S}_cg_z"] = z
    save("g6_tiny_classifier.npz", **out)


def full(seed=0):
    t0 = time.time()
    spec = synth.state_dict_spec()
    sd = synth.make_state_dict(spec, seed)
    print("weights", time.time() - t0)
    cfg = ref_import.load_ldm_config()
    model, ns = ref_import.build_reference_ldm(cfg, sd)
    check_spec(model, spec)
    print("model", time.time() - t0)
    # ---- G4: one full-size UNet forward (CFG batch of 2), inputs by seed
    x = rnd((2, 4, 16, 64), 200)
    t = torch.tensor([961, 41])
    c = rnd((2, 32, 768), 201)
    out = {}
    with torch.no_grad():
        out["unet_y"] = model.apply_model(x, t, c)
        out["unet_y_float_t"] = model.apply_model(x, torch.tensor([960.2, 40.96]), c)
        out["decode"] = model.decode_first_stage(rnd((1, 4, 16, 64), 202))[:, 0]
    save("g4_full_unet.npz", **out)
    print("g4", time.time() - t0)
    # ---- G5: full trajectories, B=1 (config 1 of BASELINE.json), seeds 21 and 22
    out = {}
    with torch.no_grad():
        for s in (21, 22):
            xT = synth.synthetic_xT(1, seed=s)
            feats = synth.synthetic_cavp(1, 32, 512, seed=1234 + s - 21)
            c = model.get_learned_conditioning(feats)
            uc = torch.zeros_like(c)
            out[f"cond_{s}"] = c[:, :2]
            z, inter = model.sample_log_diff_sampler(c, 1, "DDIM", 25, unconditional_guidance_scale=4.5,
                                                     unconditional_conditioning=uc, x_T=xT.clone())
            out[f"ddim25_z_{s}"] = z
            out[f"ddim25_mel_{s}"] = model.decode_first_stage(z)[:, 0]
            print("ddim", s, time.time() - t0)
        xT = synth.synthetic_xT(1, seed=21)
        feats = synth.synthetic_cavp(1, 32, 512, seed=1234)
        c = model.get_learned_conditioning(feats)
        uc = torch.zeros_like(c)
        # first 4 DDIM steps (short trajectory: bf16 comparisons are meaningful before chaos sets in)
        s4 = ns.DDIMSampler(model)
        s4.make_schedule(25, ddim_eta=0.0, verbose=False)
        img = xT.clone()
        steps = np.flip(s4.ddim_timesteps)
        for i in range(4):
            ts = torch.full((1,), int(steps[i]), dtype=torch.long)
            img, _ = s4.p_sample_ddim(img, c, ts, index=25 - i - 1, unconditional_guidance_scale=4.5,
                                      unconditional_conditioning=uc)
        out["ddim25_first4_x"] = img
        z, _ = model.sample_log_diff_sampler(c, 1, "DPM_Solver", 50, unconditional_guidance_scale=4.5,
                                             unconditional_conditioning=uc, x_T=xT.clone())
        out["dpm50_z_21"] = z
        out["dpm50_mel_21"] = model.decode_first_stage(z)[:, 0]
        print("dpm", time.time() - t0)
    save("g5_full_samplers.npz", **out)

    # ---- G6 full-size classifier
    cspec = synth.classifier_spec(synth.CLS_FULL)
    csd = synth.make_state_dict(cspec, seed)
    cls = ref_import.build_reference_classifier(dict(synth.CLS_FULL), csd)
    x = rnd((2, 4, 16, 64), 205)
    vf = synth.synthetic_cavp(2, 33, 512, seed=4321)
    tt = torch.tensor([500, 37])
    xin = x.clone().requires_grad_(True)
    p = cls(xin, context=vf, timesteps=tt)
    g = torch.autograd.grad(torch.log(p).sum(), xin)[0]
    save("g6_full_classifier.npz", cls_p=p, cls_grad=g)


def cavp():
    """G7: CAVP video encoder.  The reference's ResNet3dSlowOnly / CAVP_Inference code runs with the
    declared mmcv.ConvModule stand-in of oracle/ref_import.py (mmcv itself is absent): pins topology and key layout."""
    from oracle import cavp as ocavp
    CAVP_Inference = ref_import.import_reference_cavp()
    import model.cavp_modules as cm
    t0 = time.time()
    for tag, cfg, T, size in (("tiny", synth.CAVP_TINY, 4, 64), ("full", synth.CAVP_FULL, 8, 224)):
        m = CAVP_Inference("Slowonly_pool", "cnn14_pool", cfg["embed_dim"])
        if tag == "tiny":          # same classes, fewer blocks per stage
            m.video_encoder = cm.ResNet3dSlowOnly(depth=50, pretrained=None, stage_blocks=tuple(cfg["stage_blocks"]))
        m.eval()
        spec = synth.cavp_spec(cfg)
        ref = {k: tuple(v.shape) for k, v in m.state_dict().items()
               if (k.startswith("video_encoder.") or k.startswith("video_project_head.")) and "num_batches" not in k}
        assert ref == {k: tuple(v) for k, v in spec.items()}, set(ref.items()) ^ set((k, tuple(v)) for k, v in spec.items())
        sd = synth.make_state_dict(spec)
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not unexpected and all(not k.startswith("video_") or "num_batches" in k for k in missing), (missing, unexpected)
        video = synth.synthetic_video(1, T, size, seed=77)
        with torch.no_grad():
            f = m.encode_video(video, normalize=True, pool=False)
            f_raw = m.encode_video(video, normalize=False, pool=False)
        o = ocavp.encode_video(sd, video, stage_blocks=tuple(cfg["stage_blocks"]))
        err = (o - f).abs().max().item()
        print(f"cavp {tag}: feats {tuple(f.shape)}  oracle-vs-reference max|d| = {err:.2e}  ({time.time() - t0:.1f}s)")
        assert err < 1e-5
        save(f"g7_cavp_{tag}.npz", feats=f, feats_raw=f_raw)


def video_frames():
    """G9: frame pre-processing of Extract_CAVP_Features (demo_util.py:100-104, 150-151).  The per-frame transform is
    torchvision Resize + ToTensor on a PIL image, i.e. Pillow's antialiased BILINEAR resize; Pillow is installed here, so
    the fixture holds PILLOW'S OWN outputs: full uint8 results for a small case and SHA-256 digests of the results for the
    sizes the pipeline actually sees (360x640 and 1080x1920 sources -> 224x224)."""
    import hashlib
    from PIL import Image
    out = {}

    def frames(seed, T, H, W):
        rng = np.random.default_rng(seed)
        f = rng.integers(0, 256, (T, H, W, 3), dtype=np.uint8)
        yy, xx = np.mgrid[0:H, 0:W]
        f[0] = np.stack([(xx * 255 // max(W - 1, 1)), (yy * 255 // max(H - 1, 1)), ((xx + yy) % 256)], -1).astype(np.uint8)
        return f
    f = frames(900, 3, 90, 160)
    out["small_90x160_to_64x64"] = np.stack([np.asarray(Image.fromarray(x).resize((64, 64), Image.BILINEAR)) for x in f])
    for tag, (seed, T, H, W, oh, ow) in {"d360": (901, 2, 360, 640, 224, 224), "d1080": (902, 1, 1080, 1920, 224, 224),
                                         "up": (903, 2, 100, 120, 224, 224), "same": (904, 1, 224, 224, 224, 224),
                                         "tall": (905, 1, 480, 270, 224, 224)}.items():
        f = frames(seed, T, H, W)
        r = np.stack([np.asarray(Image.fromarray(x).resize((ow, oh), Image.BILINEAR)) for x in f])
        out[f"sha_{tag}"] = np.frombuffer(hashlib.sha256(r.tobytes()).digest(), dtype=np.uint8)
        out[f"spec_{tag}"] = np.array([seed, T, H, W, oh, ow])
    save("g9_video_frames.npz", **out)


def configs():
    """G8: the single-GPU BASELINE configurations that no other fixture exercises at full size.
    configs[2]: 50-step DPM-Solver++(2M) with the double-guidance classifier in the loop (CFG 4.5, classifier scale 50),
                reference run at B=1 (sample 0 of the B=8 test batch: samples are independent).
    configs[4] on one GPU: full SlowOnly-R50 on 32 frames of 224x224 -> get_learned_conditioning -> 25-step DDIM ->
                decode_first_stage, for candidates 0 and 7 of the 8 candidates per video (x_T seeded by candidate index).
    ~8 min on 8 CPU threads."""
    t0 = time.time()
    spec = synth.state_dict_spec()
    sd = synth.make_state_dict(spec, 0)
    model, ns = ref_import.build_reference_ldm(ref_import.load_ldm_config(), sd)
    cspec = synth.classifier_spec(synth.CLS_FULL)
    cls = ref_import.build_reference_classifier(dict(synth.CLS_FULL), synth.make_state_dict(cspec, 0))

    class Wrap:
        def __call__(self, x, t, video_feat):
            return cls(x, context=video_feat, timesteps=t)
    out = {}
    with torch.no_grad():
        # ---- configs[2], sample 0 of 8
        feats33 = synth.synthetic_cavp(8, 33, 512, seed=4321)[:1]
        c = model.get_learned_conditioning(feats33[:, :32])
        uc = torch.zeros_like(c)
        xT = synth.synthetic_xT(8, seed=21)[:1]
        z, _ = model.sample_log_with_classifier_diff_sampler(
            c, origin_cond=feats33, batch_size=1, sampler_name="DPM_Solver", ddim_steps=50, unconditional_guidance_scale=4.5,
            unconditional_conditioning=uc, classifier=Wrap(), classifier_guide_scale=50.0, x_T=xT.clone())
        out["c2_dpm50_cg_z0"] = z
        out["c2_dpm50_cg_mel0"] = model.decode_first_stage(z)[:, 0]
        print("configs[2]", time.time() - t0)
        # ---- configs[4] chain on one GPU
        CAVP_Inference = ref_import.import_reference_cavp()
        m = CAVP_Inference("Slowonly_pool", "cnn14_pool", synth.CAVP_FULL["embed_dim"])
        m.eval()
        missing, unexpected = m.load_state_dict(synth.make_state_dict(synth.cavp_spec(synth.CAVP_FULL)), strict=False)
        assert not unexpected
        video = synth.synthetic_video(1, 32, 224, seed=78)
        f = m.encode_video(video, normalize=True, pool=False)          # (1, 32, 512)
        out["c4_cavp_feats"] = f
        c = model.get_learned_conditioning(f)
        uc = torch.zeros_like(c)
        xT8 = synth.synthetic_xT(8, seed=21)
        for k in (0, 7):
            z, _ = model.sample_log_diff_sampler(c, 1, "DDIM", 25, unconditional_guidance_scale=4.5,
                                                 unconditional_conditioning=uc, x_T=xT8[k:k + 1].clone())
            out[f"c4_ddim25_z{k}"] = z
            out[f"c4_ddim25_mel{k}"] = model.decode_first_stage(z)[:, 0]
            print("configs[4] candidate", k, time.time() - t0)
    save("g8_full_configs.npz", **out)


def inpaint(seed=0):
    """G10: inpainting (mask / x0) through the reference samplers, tiny config: DDIM-6 and PLMS-6 with CFG 4.5, ancestral 4 steps.
    q_sample's torch.randn_like(x0) (ddpm.py:279-282) is fed from its own seeded generator so that the sequence can be replayed
    (the samplers' other noise draws stay on the global generator)."""
    spec = synth.state_dict_spec(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY)
    sd = synth.make_state_dict(spec, seed)
    cfg = ref_import.load_ldm_config(unet=synth.UNET_TINY, vae=synth.VAE_TINY, cond=synth.COND_TINY)
    model, ns = ref_import.build_reference_ldm(cfg, sd)
    B = 2
    xT = synth.synthetic_xT(B, seed=21)
    feats = synth.synthetic_cavp(B, 32, 64, seed=1234)
    x0 = rnd((B, 4, 16, 64), 301)
    mask = torch.ones(B, 1, 16, 64)
    mask[:, :, 4:12, 16:48] = 0.0                 # keep the border, regenerate the centre (ddpm.py:1478-1481)
    gq = torch.Generator().manual_seed(4242)
    real = torch.randn_like
    torch.randn_like = lambda t, **kw: torch.randn(t.shape, generator=gq)
    out = {"x0": x0, "mask": mask, "q_seed": np.int64(4242)}
    try:
        with torch.no_grad():
            c = model.get_learned_conditioning(feats)
            uc = torch.zeros_like(c)
            for name in ("DDIM", "PLMS"):
                gq.manual_seed(4242)
                z, _ = model.sample_log_diff_sampler(c, B, name, 6, unconditional_guidance_scale=4.5, unconditional_conditioning=uc,
                                                     x_T=xT.clone(), mask=mask, x0=x0)
                out[f"{name}_6_z"] = z
            gq.manual_seed(4242)
            torch.manual_seed(77)
            z, _ = model.sample(c, batch_size=B, return_intermediates=True, x_T=xT.clone(), timesteps=4, shape=(B, 4, 16, 64),
                                verbose=False, mask=mask, x0=x0)
            out["ancestral_4_z"] = z
    finally:
        torch.randn_like = real
    save("g10_tiny_inpaint.npz", **out)


def stochastic(seed=0):
    """G12: the stochastic DDIM step of the reference (ddim.py:258-273), tiny config, CFG 4.5, 6 steps: eta = 1 with temperature 1
    and 0.7, and with noise_dropout = 0.25 (torch.nn.functional.dropout on the noise, ddim.py:270-271).  Every random draw -- the
    per-step noise AND the dropout mask -- comes from the global CPU generator after torch.manual_seed(77), in the reference's
    order (noise, then mask), so a replay that draws in the same order reproduces the run."""
    spec = synth.state_dict_spec(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY)
    sd = synth.make_state_dict(spec, seed)
    cfg = ref_import.load_ldm_config(unet=synth.UNET_TINY, vae=synth.VAE_TINY, cond=synth.COND_TINY)
    model, ns = ref_import.build_reference_ldm(cfg, sd)
    B = 2
    xT = synth.synthetic_xT(B, seed=21)
    feats = synth.synthetic_cavp(B, 32, 64, seed=1234)
    out = {"noise_seed": np.int64(77)}
    with torch.no_grad():
        c = model.get_learned_conditioning(feats)
        uc = torch.zeros_like(c)
        for tag, kw in (("eta1", dict()), ("eta1_temp07", dict(temperature=0.7)), ("eta1_drop025", dict(noise_dropout=0.25)),
                        ("eta05_drop05_temp13", dict(noise_dropout=0.5, temperature=1.3))):
            torch.manual_seed(77)
            eta = 0.5 if tag.startswith("eta05") else 1.0
            z, _ = model.sample_log_diff_sampler(c, B, "DDIM", 6, unconditional_guidance_scale=4.5, unconditional_conditioning=uc,
                                                 x_T=xT.clone(), eta=eta, **kw)
            out[f"DDIM_6_{tag}_z"] = z
    save("g12_tiny_ddim_stochastic.npz", **out)


def full_extra(seed=0):
    """G5 extension (round 3): reference DDIM-25 trajectories for seeds 23 and 24, so that a full 25-step B=4 run of BASELINE
    configs[1] can be compared ROW BY ROW with four B=1 reference runs (seeds 21..24; samples are independent, so row i of
    the batch must reproduce the B=1 run of seed 21+i).  Written to its own file: g5_full_samplers.npz stays untouched."""
    t0 = time.time()
    spec = synth.state_dict_spec()
    sd = synth.make_state_dict(spec, seed)
    cfg = ref_import.load_ldm_config()
    model, ns = ref_import.build_reference_ldm(cfg, sd)
    check_spec(model, spec)
    out = {}
    with torch.no_grad():
        for s in (23, 24):
            xT = synth.synthetic_xT(1, seed=s)
            feats = synth.synthetic_cavp(1, 32, 512, seed=1234 + s - 21)
            c = model.get_learned_conditioning(feats)
            uc = torch.zeros_like(c)
            z, _ = model.sample_log_diff_sampler(c, 1, "DDIM", 25, unconditional_guidance_scale=4.5,
                                                 unconditional_conditioning=uc, x_T=xT.clone())
            out[f"ddim25_z_{s}"] = z
            out[f"ddim25_mel_{s}"] = model.decode_first_stage(z)[:, 0]
            print("ddim", s, time.time() - t0)
    save("g5_full_samplers_extra.npz", **out)


def quad():
    """G11: DDIMSampler.make_schedule(ddim_discretize="quad") of the reference (tiny model: the schedule only needs its buffers),
    S in {10, 25, 50}, eta 0 and 1."""
    spec = synth.state_dict_spec(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY)
    sd = synth.make_state_dict(spec, 0)
    cfg = ref_import.load_ldm_config(unet=synth.UNET_TINY, vae=synth.VAE_TINY, cond=synth.COND_TINY)
    model, ns = ref_import.build_reference_ldm(cfg, sd)
    out = {}
    for S in (10, 25, 50):
        for eta in (0.0, 1.0):
            s = ns.DDIMSampler(model)
            s.make_schedule(S, ddim_discretize="quad", ddim_eta=eta, verbose=False)
            tag = f"quad{S}_eta{int(eta)}"
            out[f"{tag}_timesteps"] = s.ddim_timesteps
            out[f"{tag}_alphas"] = np.asarray(s.ddim_alphas, dtype=np.float64)
            out[f"{tag}_alphas_prev"] = np.asarray(s.ddim_alphas_prev, dtype=np.float64)
            out[f"{tag}_sqrt_one_minus_alphas"] = np.asarray(s.ddim_sqrt_one_minus_alphas, dtype=np.float64)
            out[f"{tag}_sigmas"] = np.asarray(s.ddim_sigmas, dtype=np.float64)
    save("g11_ddim_quad.npz", **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--cavp", action="store_true", help="CAVP video encoder vectors (reference topology, mmcv stand-in)")
    ap.add_argument("--video", action="store_true", help="G9: frame pre-processing vectors (Pillow's own resize outputs)")
    ap.add_argument("--configs", action="store_true", help="G8: BASELINE configs[2] / configs[4] at full size (~8 min)")
    ap.add_argument("--inpaint", action="store_true", help="G10: mask / x0 inpainting through DDIM, PLMS and the ancestral sampler (tiny)")
    ap.add_argument("--quad", action="store_true", help="G11: the 'quad' DDIM discretisation tables (seconds)")
    ap.add_argument("--stochastic", action="store_true", help="G12: eta > 0 DDIM with temperature / noise_dropout (tiny, seconds)")
    ap.add_argument("--full-extra", action="store_true", help="G5 extension: DDIM-25 reference runs for seeds 23 / 24 (~4 min)")
    a = ap.parse_args()
    torch.set_num_threads(8)
    if a.tiny:
        tiny()
    if a.full:
        full()
    if a.cavp:
        cavp()
    if a.video:
        video_frames()
    if a.configs:
        configs()
    if a.full_extra:
        full_extra()
    if a.inpaint:
        inpaint()
    if a.quad:
        quad()
    if a.stochastic:
        stochastic()
