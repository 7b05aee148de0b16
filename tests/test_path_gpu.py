"""GPU parity tests of the hot path through the C ABI (LatentDiffusion facade -> libdfengine.so) against
(i) the CPU oracle on the same seeded inputs and (ii) the golden vectors produced by the reference itself.

Tolerances (bf16 MFMA operands, fp32 accumulation / statistics / residual stream; oracle and reference are fp32):
  * one UNet / VAE forward:         rel-L2 <= 2e-2  (measured 1.2e-2 / 7e-3 at full size)
  * sampler trajectories:           rel-L2 <= 5e-2  (measured 5e-3 .. 7e-3 over 25-50 full-size steps)
  * decoded mel, north-star metric: MAE < 1e-3 in units of the reference mel's range (measured 8.8e-4), and
                                    < 1.5 % of its standard deviation in absolute terms (test_full_ddim25_mel_mae)
"""
import numpy as np
import pytest
import torch

from helpers import (gold, rnd, rel_l2, tiny_state_dict, full_state_dict, tiny_classifier_sd, full_classifier_sd)

pytestmark = pytest.mark.gpu

FWD_TOL = 2e-2
TRAJ_TOL = 5e-2


@pytest.fixture(scope="module")
def P():
    import diff_foley_amd
    return diff_foley_amd


@pytest.fixture(scope="module")
def tiny(P):
    from diff_foley_amd import synth
    # this file pins the bf16-operand build (BASELINE configs[1]'s literal "bf16 UNet"); the facade's DEFAULT is the fp16
    # build, the one that meets the north-star tolerance -- its parity tests are tests/test_path_fp16_gpu.py
    m = P.LatentDiffusion(precision="bf16", **P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY))
    m.load_state_dict(tiny_state_dict())
    m.cuda()
    return m


@pytest.fixture(scope="module")
def full(P):
    m = P.LatentDiffusion(precision="bf16", **P.stage2_config())
    m.load_state_dict(full_state_dict())
    m.cuda()
    return m


def _oracle_tiny():
    from diff_foley_amd import synth
    from oracle import unet as ou
    sd = tiny_state_dict()
    usd = ou.sub_state_dict(sd, "model.diffusion_model.")
    vsd = ou.sub_state_dict(sd, "first_stage_model.")
    csd = ou.sub_state_dict(sd, "cond_stage_model.")
    return usd, vsd, csd, synth


# ------------------------------------------------------------------------------------------- tiny config
def test_tiny_unet_vs_golden_and_oracle(tiny):
    g = gold("g3_tiny_unet.npz")
    x, c = rnd((2, 4, 16, 64), 102), rnd((2, 32, 128), 101)
    y = tiny.apply_model(x.cuda(), torch.tensor([500, 37]).cuda(), c.cuda()).cpu()
    assert y.shape == g["y_int"].shape and y.dtype == torch.float32
    assert rel_l2(y, g["y_int"]) < FWD_TOL
    y = tiny.apply_model(x.cuda(), torch.tensor([500.25, 37.7]).cuda(), c.cuda()).cpu()     # fractional t (DPM path)
    assert rel_l2(y, g["y_flt"]) < FWD_TOL


def test_tiny_small_latent_ops_golden(tiny):
    """G3: reference tensors captured on an 8x16 latent (other plan shapes than 16x64)."""
    g = gold("g3_tiny_ops.npz")
    y = tiny.apply_model(g["x"].cuda(), g["t"].cuda(), g["c"].cuda()).cpu()
    assert rel_l2(y, g["y"]) < FWD_TOL


def test_tiny_per_block_vs_reference_tensors(tiny):
    """G3 per block: the tensors the reference's forward hooks captured at the input and output of individual ResBlocks
    (with / without skip conv, three resolutions), SpatialTransformers (64-, 8- and 2-token maps), a Downsample and an
    Upsample, fed through the plan builder's code path for exactly that block (df_test_unet_block).  A regression in
    one block shows up here under its own name instead of only in the end-to-end rel-L2."""
    import torch.nn.functional as F
    g = gold("g3_tiny_ops.npz")
    eng = tiny.engine
    worst = {}
    for p in ("input_blocks.1.0", "input_blocks.4.0", "output_blocks.5.0"):
        y = eng.test_block(p, 0, g[p + "__in"], semb=F.silu(g[p + "__in1"]), cout=g[p + "__out"].shape[1]).cpu()
        worst[p] = rel_l2(y, g[p + "__out"])
    for p in ("input_blocks.1.1", "middle_block.1", "output_blocks.5.1"):
        y = eng.test_block(p, 1, g[p + "__in"], context=g[p + "__in1"]).cpu()
        worst[p] = rel_l2(y, g[p + "__out"])
    y = eng.test_block("input_blocks.3.0", 2, g["input_blocks.3.0__in"]).cpu()
    worst["input_blocks.3.0 (down)"] = rel_l2(y, g["input_blocks.3.0__out"])
    y = eng.test_block("output_blocks.2.1", 3, g["output_blocks.2.1__in"]).cpu()
    worst["output_blocks.2.1 (up)"] = rel_l2(y, g["output_blocks.2.1__out"])
    print("per-block rel-L2:", {k: f"{v:.2e}" for k, v in worst.items()})
    for k, v in worst.items():
        assert v < 1e-2, (k, v)          # one block, bf16 operands: well inside the whole-UNet tolerance of 2e-2


def test_tiny_vae_and_cond_vs_golden(tiny):
    g = gold("g3_tiny_unet.npz")
    d = tiny.decode_first_stage(rnd((2, 4, 16, 64), 103).cuda()).cpu()
    assert d.shape == g["decode"].shape
    assert rel_l2(d, g["decode"]) < FWD_TOL
    c = tiny.get_learned_conditioning(rnd((2, 32, 64), 104).cuda()).cpu()
    assert rel_l2(c, g["cond"]) < 5e-3


def test_tiny_batch_sizes_and_cfg_forward(tiny):
    """Batch 1 / 3 / 8 plans and the fused CFG entry point against the oracle."""
    usd, _, _, synth = _oracle_tiny()
    from oracle import unet as ou
    for B in (1, 3, 8):
        x, c = rnd((B, 4, 16, 64), 300 + B), rnd((B, 32, 128), 400 + B)
        t = torch.arange(B) * 100 + 7
        ref = ou.unet_forward(usd, synth.UNET_TINY, x, t, c)
        y = tiny.apply_model(x.cuda(), t.cuda(), c.cuda()).cpu()
        assert rel_l2(y, ref) < FWD_TOL, B
    B = 2
    x, c = rnd((B, 4, 16, 64), 310), rnd((B, 32, 128), 410)
    uc = torch.zeros_like(c)
    t = torch.tensor([961, 961])
    e2 = ou.unet_forward(usd, synth.UNET_TINY, torch.cat([x, x]), torch.cat([t, t]), torch.cat([uc, c]))
    ref = e2[:B] + 4.5 * (e2[B:] - e2[:B])
    tiny.engine.set_context(torch.cat([uc, c]).cuda())
    y = tiny.engine.unet_forward_cfg(x.cuda(), t.float().cuda(), 4.5).cpu()
    assert rel_l2(y, ref) < TRAJ_TOL        # guidance amplifies the (e_c - e_u) rounding error by 4.5x


def test_tiny_samplers_vs_golden(tiny):
    g = gold("g5_tiny_samplers.npz")
    from diff_foley_amd import synth
    B = 2
    xT = synth.synthetic_xT(B, seed=21)
    c = tiny.get_learned_conditioning(synth.synthetic_cavp(B, 32, 64, seed=1234).cuda())
    uc = torch.zeros_like(c)
    for name, S in (("DDIM", 25), ("DPM_Solver", 25), ("DPM_Solver", 10), ("PLMS", 25)):
        z, inter = tiny.sample_log_diff_sampler(c, B, name, S, unconditional_guidance_scale=4.5,
                                                unconditional_conditioning=uc, x_T=xT.clone())
        assert z.shape == (B, 4, 16, 64) and z.dtype == torch.float32
        err = rel_l2(z.cpu(), g[f"{name}_{S}_z"])
        assert err < TRAJ_TOL, (name, S, err)
        if name == "DPM_Solver":
            assert inter is None
        else:
            assert len(inter["x_inter"]) == int(g[f"{name}_{S}_n_inter"]) and len(inter["pred_x0"]) == len(inter["x_inter"])
            assert rel_l2(inter["pred_x0"][-1].cpu(), g[f"{name}_{S}_pred_x0_last"]) < TRAJ_TOL
    z, _ = tiny.sample_log_diff_sampler(c, B, "DDIM", 25, x_T=xT.clone())
    assert rel_l2(z.cpu(), g["DDIM_25_nocfg_z"]) < TRAJ_TOL
    z, _ = tiny.sample_log_diff_sampler(c, B, "DDIM", 25, unconditional_guidance_scale=4.5,
                                        unconditional_conditioning=uc, x_T=xT.clone())
    mel = tiny.decode_first_stage(z).cpu()
    assert rel_l2(mel, g["DDIM_25_mel"]) < TRAJ_TOL


def test_tiny_ancestral_sampler_vs_golden(tiny):
    g = gold("g5_tiny_samplers.npz")
    from diff_foley_amd import synth
    B = 2
    xT = synth.synthetic_xT(B, seed=21)
    c = tiny.get_learned_conditioning(synth.synthetic_cavp(B, 32, 64, seed=1234).cuda())
    torch.manual_seed(77)                      # the golden run drew its noise from the CPU generator
    z, inter = tiny.sample(c, batch_size=B, return_intermediates=True, x_T=xT.clone(), timesteps=6,
                           shape=(B, 4, 16, 64), noise_fn=lambda s: torch.randn(s))
    assert rel_l2(z.cpu(), g["ancestral_6_z"]) < TRAJ_TOL


def test_tiny_inpainting_vs_golden(tiny):
    """mask / x0 (inpainting) through the product samplers against the reference's own runs (G10, tests/golden/make_golden.py
    --inpaint): DDIM and PLMS with CFG, the ancestral sampler, and the DPM-Solver sampler refusing what its reference drops.
    The blend runs in df_q_sample_blend (csrc/elementwise.hip); q_sample's noise is replayed through the q_noise_fn hook."""
    g = gold("g10_tiny_inpaint.npz")
    from diff_foley_amd import synth
    B = 2
    xT = synth.synthetic_xT(B, seed=21)
    c = tiny.get_learned_conditioning(synth.synthetic_cavp(B, 32, 64, seed=1234).cuda())
    uc = torch.zeros_like(c)
    x0, mask = g["x0"], g["mask"]
    gq = torch.Generator()
    qn = lambda shape: torch.randn(tuple(shape), generator=gq)
    for name in ("DDIM", "PLMS"):
        gq.manual_seed(int(g["q_seed"]))
        z, _ = tiny.sample_log_diff_sampler(c, B, name, 6, unconditional_guidance_scale=4.5, unconditional_conditioning=uc,
                                            x_T=xT.clone(), mask=mask, x0=x0, q_noise_fn=qn)
        assert rel_l2(z.cpu(), g[f"{name}_6_z"]) < TRAJ_TOL, name
    gq.manual_seed(int(g["q_seed"]))
    torch.manual_seed(77)
    z, _ = tiny.sample(c, batch_size=B, return_intermediates=True, x_T=xT.clone(), timesteps=4, shape=(B, 4, 16, 64),
                       noise_fn=lambda s: torch.randn(s), mask=mask, x0=x0, q_noise_fn=qn)
    assert rel_l2(z.cpu(), g["ancestral_4_z"]) < TRAJ_TOL
    # a [B][C][H][W] mask gives the same result as the broadcast [B][1][H][W] one; without a hook the noise is drawn on the device
    gq.manual_seed(int(g["q_seed"]))
    z2, _ = tiny.sample_log_diff_sampler(c, B, "DDIM", 6, unconditional_guidance_scale=4.5, unconditional_conditioning=uc,
                                         x_T=xT.clone(), mask=mask.expand(B, 4, 16, 64), x0=x0, q_noise_fn=qn)
    assert rel_l2(z2.cpu(), g["DDIM_6_z"]) < TRAJ_TOL
    z3, _ = tiny.sample_log_diff_sampler(c, B, "DDIM", 6, x_T=xT.clone(), mask=mask, x0=x0)
    assert torch.isfinite(z3).all()
    with pytest.raises(NotImplementedError):
        tiny.sample_log_diff_sampler(c, B, "DPM_Solver", 6, x_T=xT.clone(), mask=mask, x0=x0)
    with pytest.raises(AssertionError):
        tiny.sample_log_diff_sampler(c, B, "DDIM", 6, x_T=xT.clone(), mask=mask)          # mask without x0: like the reference


def test_tiny_stochastic_ddim_vs_golden(tiny):
    """eta > 0 DDIM with temperature and noise_dropout (ddim.py:269-271) through the product sampler against the reference's own
    runs (G12).  The reference drew noise and dropout mask from the global CPU generator; the `noise_fn` hook hands the sampler CPU
    noise from the same generator, and the sampler applies the dropout to that tensor where it lives, in the reference's order."""
    g = gold("g12_tiny_ddim_stochastic.npz")
    from diff_foley_amd import synth
    B = 2
    xT = synth.synthetic_xT(B, seed=21)
    c = tiny.get_learned_conditioning(synth.synthetic_cavp(B, 32, 64, seed=1234).cuda())
    uc = torch.zeros_like(c)
    for tag, eta, kw in (("eta1", 1.0, dict()), ("eta1_temp07", 1.0, dict(temperature=0.7)),
                         ("eta1_drop025", 1.0, dict(noise_dropout=0.25)),
                         ("eta05_drop05_temp13", 0.5, dict(noise_dropout=0.5, temperature=1.3))):
        torch.manual_seed(int(g["noise_seed"]))
        z, _ = tiny.sample_log_diff_sampler(c, B, "DDIM", 6, unconditional_guidance_scale=4.5, unconditional_conditioning=uc,
                                            x_T=xT.clone(), eta=eta, noise_fn=lambda s: torch.randn(s), **kw)
        assert rel_l2(z.cpu(), g[f"DDIM_6_{tag}_z"]) < TRAJ_TOL, tag
    # without the hook the noise comes from the device generator: finite, and different from the eta = 0 trajectory
    z2, _ = tiny.sample_log_diff_sampler(c, B, "DDIM", 6, x_T=xT.clone(), eta=1.0, noise_dropout=0.25)
    z0, _ = tiny.sample_log_diff_sampler(c, B, "DDIM", 6, x_T=xT.clone())
    assert torch.isfinite(z2).all() and rel_l2(z2.cpu(), z0.cpu()) > 1e-2
    # PLMS (eta = 0: no noise to drop) and DPM-Solver++ (its reference sampler drops the argument) accept it like the reference
    for name in ("PLMS", "DPM_Solver"):
        za, _ = tiny.sample_log_diff_sampler(c, B, name, 6, x_T=xT.clone(), noise_dropout=0.25)
        zb, _ = tiny.sample_log_diff_sampler(c, B, name, 6, x_T=xT.clone())
        assert torch.equal(za, zb), name


def test_score_corrector_callback(P, tiny):
    """score_corrector.modify_score(model, e_t, x, t, c, **corrector_kwargs) (ddim.py:249-251, 382-384, plms.py:186-188): a caller's
    callback on the guided eps.  An identity corrector changes nothing; a scaling one gives the trajectory of the scaled eps
    (checked against the oracle's DDIM loop driven by the product's own eps * k); `t` arrives as the reference's int64 `ts`; a
    corrector that evaluates the model itself (apply_model under ANOTHER conditioning: the engine's context changes hands) leaves
    the sampler's own [uncond ; cond] context intact for the next step; with classifier guidance the callback sees the eps AFTER
    the classifier gradient, the reference's order (ddim.py:374-384)."""
    from diff_foley_amd import synth
    B = 2
    xT = synth.synthetic_xT(B, seed=21)
    c = tiny.get_learned_conditioning(synth.synthetic_cavp(B, 32, 64, seed=1234).cuda())
    uc = torch.zeros_like(c)
    calls = []

    class Corr:
        def __init__(self, k):
            self.k = k

        def modify_score(self, model, e_t, x, t, cond, gain=1.0):
            assert t.dtype == torch.int64 and t.shape == (B,)
            model.alphas_cumprod[t]                    # schedule tables are indexed with it (extract_into_tensor -> gather)
            calls.append((tuple(e_t.shape), float(t[0]), gain))
            assert model is tiny and cond is c
            return e_t * (self.k * gain)
    kw = dict(unconditional_guidance_scale=4.5, unconditional_conditioning=uc)
    for name in ("DDIM", "PLMS"):
        z0, _ = tiny.sample_log_diff_sampler(c, B, name, 6, x_T=xT.clone(), **kw)
        calls.clear()
        z1, _ = tiny.sample_log_diff_sampler(c, B, name, 6, x_T=xT.clone(), score_corrector=Corr(1.0), **kw)
        assert torch.equal(z0, z1) and len(calls) == (7 if name == "DDIM" else 8)      # S = 6 -> 7 steps; PLMS: +1 eps at step 0
    calls.clear()
    z2, _ = tiny.sample_log_diff_sampler(c, B, "DDIM", 6, x_T=xT.clone(), score_corrector=Corr(0.5),
                                         corrector_kwargs=dict(gain=1.5), **kw)
    assert all(g == 1.5 for _, _, g in calls)
    from oracle import samplers as osamp, schedule as osch
    # the oracle's loop driven by the product's own GUIDED eps * k, taken from the plan the sampler itself runs (the 2B-row CFG
    # plan: the tiny random-weight bf16 model turns ANY rounding-level difference between two plans -- another tile, another
    # split -- into ~1e-2 after 6 steps, tools/sens_probe.py; what is checked here is the callback, not plan-to-plan agreement)
    tiny.engine.set_context(torch.cat([uc, c]))
    eps = lambda x, t, cc: tiny.engine.unet_forward_cfg(x.cuda(), t.cuda().float(), 4.5).cpu()
    zo, _ = osamp.ddim_sample(lambda x, t, cc: 0.75 * eps(x, t, cc), osch.ddpm_schedule()["alphas_cumprod"], 6, xT, c.cpu())
    assert rel_l2(z2.cpu(), zo) < 5e-3        # (a wrong factor -- 0.5 or 1.0 for 0.75 -- is an O(1) difference)
    # ... and one check that does NOT share the engine's eps: the oracle's fp32 UNet (oracle/unet.py) through the oracle's CFG loop,
    # its eps scaled by the same 0.75 (the guidance combination is linear in eps, so scaling the model's output scales the guided
    # eps).  Tolerance: TRAJ_TOL, what this tiny bf16 model's 6-step trajectories are held to against the reference's goldens.
    usd, vsd, csd, _ = _oracle_tiny()
    from oracle import unet as ou, vae as ov
    c_ref = ov.cond_stage(csd, synth.synthetic_cavp(B, 32, 64, seed=1234))
    zi, _ = osamp.ddim_sample(lambda x, t, cc: 0.75 * ou.unet_forward(usd, synth.UNET_TINY, x, t, cc),
                              osch.ddpm_schedule()["alphas_cumprod"], 6, xT, c_ref, scale=4.5, uc=torch.zeros_like(c_ref))
    assert rel_l2(z2.cpu(), zi) < TRAJ_TOL

    # a corrector that calls the model: apply_model and the module facades put THEIR context into the engine (B rows, another
    # conditioning); the next step's CFG forward needs the sampler's 2B-row context back
    other = tiny.get_learned_conditioning(synth.synthetic_cavp(B, 32, 64, seed=77).cuda())
    seen = []

    class CallsModel:
        def modify_score(self, model, e_t, x, t, cond):
            e_other = model.apply_model(x, t, other)
            e_fac = model.model.diffusion_model(x, t.float(), context=other)
            seen.append(float((e_other - e_fac).abs().max()))
            return e_t + 0.0 * e_other
    for name in ("DDIM", "PLMS"):
        z0, _ = tiny.sample_log_diff_sampler(c, B, name, 6, x_T=xT.clone(), **kw)
        z3, _ = tiny.sample_log_diff_sampler(c, B, name, 6, x_T=xT.clone(), score_corrector=CallsModel(), **kw)
        assert torch.equal(z0, z3), name
        z4, _ = tiny.sample_log_diff_sampler(c, B, name, 6, x_T=xT.clone(), score_corrector=CallsModel())   # no CFG: same rule
        z5, _ = tiny.sample_log_diff_sampler(c, B, name, 6, x_T=xT.clone())
        assert torch.equal(z4, z5), name
    assert seen and max(seen) == 0.0

    # classifier guidance + corrector: eps' = corrector(eps_cfg - sqrt(1 - a_t) * s * grad)   (ddim.py:374-384)
    cls = P.AlignmentClassifier(classifier_config=dict(params=dict(synth.CLS_TINY)))
    cls.load_state_dict(tiny_classifier_sd())
    cls.attach(tiny)
    vf = synth.synthetic_cavp(B, 33, 64, seed=4321).cuda()
    got = []

    class Record:
        def modify_score(self, model, e_t, x, t, cond):
            got.append(e_t.clone())
            return e_t
    ckw = dict(origin_cond=vf, batch_size=B, sampler_name="DDIM", ddim_steps=4, classifier=cls, classifier_guide_scale=50.0,
               x_T=xT.clone(), **kw)
    za, _ = tiny.sample_log_with_classifier_diff_sampler(c, **ckw)
    zb, _ = tiny.sample_log_with_classifier_diff_sampler(c, score_corrector=Record(), **ckw)
    from diff_foley_amd.schedule import DDIMTables
    tb = DDIMTables(tiny.alphas_cumprod, 4)
    assert torch.equal(za, zb) and len(got) == len(tb.timesteps)
    # first step: the recorded eps is the CFG eps minus the classifier term, not the bare CFG eps
    tiny.engine.set_context(torch.cat([uc, c]))
    t0 = torch.full((B,), float(np.flip(tb.timesteps)[0]), device="cuda")
    e_cfg = tiny.engine.unet_forward_cfg(xT.cuda(), t0, 4.5)
    grad = cls.log_prob_grad(xT.cuda(), t0, vf)
    want = e_cfg - float(np.sqrt(1.0 - tb.alphas[len(tb.alphas) - 1])) * 50.0 * grad
    assert rel_l2(got[0].cpu(), want.cpu()) < 1e-5 and rel_l2(got[0].cpu(), e_cfg.cpu()) > 1e-4


def test_classifier_gradient_on_the_side_stream_is_bit_identical(tiny, P, monkeypatch):
    """Round 6: the classifier gradient of a guided step runs on a second HIP stream beside the UNet call (samplers.py
    _eps_and_classifier_grad).  Same latents, bit for bit, as the one-stream order -- DDIM (ddim.py:374-380) and DPM-Solver++
    double guidance (dpm_solver.py:1377-1393), repeated so that a missing stream dependency would get its chances."""
    from diff_foley_amd import synth
    B = 2
    c = tiny.get_learned_conditioning(synth.synthetic_cavp(B, 32, 64, seed=1234).cuda())
    uc = torch.zeros_like(c)
    xT = synth.synthetic_xT(B, seed=21).cuda()
    cls = P.AlignmentClassifier(classifier_config=dict(params=dict(synth.CLS_TINY)))
    cls.load_state_dict(tiny_classifier_sd())
    cls.attach(tiny)
    vf = synth.synthetic_cavp(B, 33, 64, seed=4321).cuda()
    for name, steps in (("DDIM", 6), ("DPM_Solver", 8)):
        kw = dict(origin_cond=vf, batch_size=B, sampler_name=name, ddim_steps=steps, unconditional_guidance_scale=4.5,
                  unconditional_conditioning=uc, classifier=cls, classifier_guide_scale=50.0)
        monkeypatch.setenv("DF_CLS_OVERLAP", "0")
        z0, _ = tiny.sample_log_with_classifier_diff_sampler(c, x_T=xT.clone(), **kw)
        monkeypatch.setenv("DF_CLS_OVERLAP", "1")
        for rep in range(4):
            z1, _ = tiny.sample_log_with_classifier_diff_sampler(c, x_T=xT.clone(), **kw)
            assert torch.isfinite(z1).all() and torch.equal(z0, z1), (name, rep)
        monkeypatch.delenv("DF_CLS_OVERLAP")
        z2, _ = tiny.sample_log_with_classifier_diff_sampler(c, x_T=xT.clone(), **kw)       # the default is the side stream
        assert torch.equal(z0, z2), name


def test_classifier_feature_operands_are_reused_across_a_guidance_loop(tiny, P, monkeypatch):
    """df_classifier_grad_cached: a guidance loop passes the SAME video features at every step (origin_cond, ddim.py:374-380,
    dpm_solver.py:1377-1393), so the feature-only launches of the gradient plan (cast + cross-attention K / V^T per transformer
    block) run once per features, not once per step.  A reused call returns the bits of a recomputing call; a new tensor object, an
    in-place edit (version counter) or a rebuilt plan recomputes; an edit torch cannot see (``.data``) is NOT picked up -- the same
    rule as the UNet's context cache -- which also shows that the reuse really skips the feature launches; DF_CLS_FEAT_CACHE=0 is
    the reference's recompute-every-call behaviour; the samplers give the same latents either way."""
    from diff_foley_amd import synth
    B = 2
    cls = P.AlignmentClassifier(classifier_config=dict(params=dict(synth.CLS_TINY)))
    cls.load_state_dict(tiny_classifier_sd())
    cls.attach(tiny)
    eng = cls.engine
    vf = synth.synthetic_cavp(B, 33, 64, seed=4321).cuda()
    xs = [synth.synthetic_xT(B, seed=30 + k).cuda() for k in range(3)]
    ts = [torch.tensor([900.0, 900.0]).cuda(), torch.tensor([500.5, 37.0]).cuda(), torch.tensor([1.0, 999.0]).cuda()]
    fresh = [eng.classifier_grad(x, t, vf.clone(), want_prob=True) for x, t in zip(xs, ts)]       # a new object every call: recomputed
    for rep in range(2):
        for k, (x, t) in enumerate(zip(xs, ts)):                                                  # one object: computed once
            g, p = eng.classifier_grad(x, t, vf, want_prob=True)
            assert torch.equal(g, fresh[k][0]) and torch.equal(p, fresh[k][1]), (rep, k)
    # an in-place edit bumps the version counter: recomputed
    vf2 = vf.clone()
    g_a = eng.classifier_grad(xs[0], ts[0], vf2)
    vf2.mul_(-1.0)
    g_b = eng.classifier_grad(xs[0], ts[0], vf2)
    g_b_want = eng.classifier_grad(xs[0], ts[0], (-vf).contiguous())
    assert torch.equal(g_b, g_b_want) and not torch.equal(g_a, g_b)
    # an edit behind torch's back is not seen (documented): the K / V^T operands of the previous call are still in use
    vf3 = vf.clone()
    g_c = eng.classifier_grad(xs[1], ts[1], vf3)
    vf3.data.mul_(-1.0)
    assert torch.equal(eng.classifier_grad(xs[1], ts[1], vf3), g_c)
    monkeypatch.setenv("DF_CLS_FEAT_CACHE", "0")
    assert not torch.equal(eng.classifier_grad(xs[1], ts[1], vf3), g_c)
    monkeypatch.delenv("DF_CLS_FEAT_CACHE")
    # a rebuilt plan starts without operands whatever token it is handed
    g_d = eng.classifier_grad(xs[2], ts[2], vf)
    eng.finalize()
    assert torch.equal(eng.classifier_grad(xs[2], ts[2], vf), g_d)
    # another batch = another plan with its own operands, interleaved with the first
    vf1 = vf[:1].contiguous()
    g1 = eng.classifier_grad(xs[0][:1].contiguous(), ts[0][:1], vf1)
    assert torch.equal(eng.classifier_grad(xs[2], ts[2], vf), g_d)
    assert torch.equal(eng.classifier_grad(xs[0][:1].contiguous(), ts[0][:1], vf1), g1)
    assert torch.equal(g1, eng.classifier_grad(xs[0][:1].contiguous(), ts[0][:1], vf1.clone()))
    # through the samplers
    c = tiny.get_learned_conditioning(synth.synthetic_cavp(B, 32, 64, seed=1234).cuda())
    kw = dict(origin_cond=vf, batch_size=B, sampler_name="DDIM", ddim_steps=5, unconditional_guidance_scale=4.5,
              unconditional_conditioning=torch.zeros_like(c), classifier=cls, classifier_guide_scale=50.0)
    xT = synth.synthetic_xT(B, seed=21).cuda()
    z1, _ = tiny.sample_log_with_classifier_diff_sampler(c, x_T=xT.clone(), **kw)
    monkeypatch.setenv("DF_CLS_FEAT_CACHE", "0")
    z0, _ = tiny.sample_log_with_classifier_diff_sampler(c, x_T=xT.clone(), **kw)
    assert torch.isfinite(z1).all() and torch.equal(z0, z1)


def test_hoisted_time_embedding_is_bit_identical(tiny):
    """df_unet_set_timesteps + df_unet_forward(_cfg)_ts: the time embedding of every announced timestep (integer and fractional,
    as DDIM / DPM-Solver++ visit them) computed before the loop by the plan's own ops; a step that looks its row up returns
    exactly what the in-step form returns, the samplers give the same latents with and without the hoist, and an index
    outside the table or a plan without a table is an error, not a silent fallback."""
    from diff_foley_amd import synth, samplers
    B = 2
    eng = tiny.engine
    eng.finalize()               # drop the plans (and timestep tables) earlier tests of this session left: the "no table yet" case below
    x = synth.synthetic_xT(B, seed=5).cuda()
    c = tiny.get_learned_conditioning(synth.synthetic_cavp(B, 32, 64, seed=1234).cuda())
    uc = torch.zeros_like(c)
    ts = [961.0, 41.0, 1.0, 960.2, 40.96, 999.0]
    eng.set_context(torch.cat([uc, c]))
    eng.set_timesteps(ts, B, 16, 64, True)
    for k, t in enumerate(ts):
        a = eng.unet_forward_cfg(x, torch.full((B,), t, device="cuda"), 4.5)
        b = eng.unet_forward_cfg(x, None, 4.5, ts_index=k)
        assert torch.equal(a, b), (k, t)
    with pytest.raises(RuntimeError):
        eng.unet_forward_cfg(x, None, 4.5, ts_index=len(ts))
    # round 6: announcing the timesteps the table already holds is a no-op; other timesteps of the same count rebuild it
    ref3 = eng.unet_forward_cfg(x, None, 4.5, ts_index=3)
    eng.set_timesteps(ts, B, 16, 64, True)
    assert torch.equal(eng.unet_forward_cfg(x, None, 4.5, ts_index=3), ref3)
    ts2 = [t + 7.0 if t < 900 else t - 7.0 for t in ts]
    eng.set_timesteps(ts2, B, 16, 64, True)
    got = eng.unet_forward_cfg(x, None, 4.5, ts_index=3)
    assert torch.equal(got, eng.unet_forward_cfg(x, torch.full((B,), ts2[3], device="cuda"), 4.5)) and not torch.equal(got, ref3)
    eng.set_timesteps(ts, B, 16, 64, True)
    assert torch.equal(eng.unet_forward_cfg(x, None, 4.5, ts_index=3), ref3)
    eng.set_context(c)
    with pytest.raises(RuntimeError):                       # the non-CFG plan of this shape has no table yet
        eng.unet_forward(x, None, ts_index=0)
    eng.set_timesteps(ts[:2], B, 16, 64, False)
    assert torch.equal(eng.unet_forward(x, torch.full((B,), ts[1], device="cuda")), eng.unet_forward(x, None, ts_index=1))
    kw = dict(unconditional_guidance_scale=4.5, unconditional_conditioning=uc)
    for name, S in (("DDIM", 6), ("PLMS", 6), ("DPM_Solver", 6)):
        z_h, _ = tiny.sample_log_diff_sampler(c, B, name, S, x_T=x.clone(), **kw)
        samplers._NO_HOIST = True
        try:
            z_i, _ = tiny.sample_log_diff_sampler(c, B, name, S, x_T=x.clone(), **kw)
        finally:
            samplers._NO_HOIST = False
        assert torch.equal(z_h, z_i), name


def test_module_facades_run_the_engine(tiny):
    """Code outside the samplers reaches the sub-modules by name (ddim.py:18, 186; ddpm.py:568-579, 739-797, 1552-1571):
    model.model.diffusion_model(x, t, context=c), model.model(x, t, c_crossattn=[c]), first_stage_model.decode(z),
    cond_stage_model(feats) run on the engine and agree with apply_model / decode_first_stage / get_learned_conditioning;
    the module tree itself (parameters, weights) is not there and says so."""
    x, c = rnd((2, 4, 16, 64), 102).cuda(), rnd((2, 32, 128), 101).cuda()
    t = torch.tensor([500, 37]).cuda()
    y = tiny.apply_model(x, t, c)
    assert torch.equal(tiny.model.diffusion_model(x, t, context=c), y)
    assert torch.equal(tiny.model(x, t, c_crossattn=[c]), y)
    assert tiny.model.conditioning_key == "crossattn"
    z = rnd((1, 4, 16, 64), 103).cuda()
    want = tiny.decode_first_stage(z)
    got = tiny.first_stage_model.decode(z / tiny.scale_factor)
    assert rel_l2(got.cpu(), want.cpu()) < 1e-5
    feats = rnd((2, 32, 64), 104).cuda()
    assert torch.equal(tiny.cond_stage_model(feats), tiny.get_learned_conditioning(feats))
    assert torch.equal(tiny.cond_stage_model.encode(feats), tiny.get_learned_conditioning(feats))
    with pytest.raises(AttributeError, match="not materialised"):
        tiny.model.diffusion_model.input_blocks
    with pytest.raises(AttributeError, match="not materialised"):
        tiny.first_stage_model.decoder
    with pytest.raises(NotImplementedError):
        tiny.first_stage_model.encode(z)


def test_tiny_ddim_eta_and_intermediates_shape(tiny):
    """eta > 0 draws noise on the device; only shapes/finite-ness and the log_every_t bookkeeping are checked."""
    from diff_foley_amd import synth
    B = 2
    c = tiny.get_learned_conditioning(synth.synthetic_cavp(B, 32, 64).cuda())
    z, inter = tiny.sample_log_diff_sampler(c, B, "DDIM", 50, eta=1.0, log_every_t=10)
    assert torch.isfinite(z).all()
    assert len(inter["x_inter"]) == 1 + 5 + 0 + (1 if 49 % 10 != 0 else 0)


def test_tiny_classifier_forward_vs_golden(P, tiny):
    from diff_foley_amd import synth
    g = gold("g6_tiny_classifier.npz")
    cls = P.AlignmentClassifier(classifier_config=dict(params=dict(synth.CLS_TINY)))
    cls.load_state_dict(tiny_classifier_sd())
    cls.attach(tiny)
    x = rnd((2, 4, 16, 64), 105)
    vf = synth.synthetic_cavp(2, 33, 64, seed=4321)
    p = cls(x.cuda(), t=torch.tensor([500.0, 37.0]).cuda(), video_feat=vf.cuda()).cpu()      # keywords, as ddim.py:338 calls it
    assert p.shape == (2, 1)
    assert torch.allclose(p, g["cls_p"], atol=2e-2)


def test_tiny_classifier_grad_and_double_guidance_vs_golden(P, tiny):
    """G6: input gradient of log p (native backward pass) and the two double-guidance samplers (ddim.py:344-396,
    dpm_solver.py:1377-1393) against the reference's autograd run."""
    from diff_foley_amd import synth
    g = gold("g6_tiny_classifier.npz")
    cls = P.AlignmentClassifier(classifier_config=dict(params=dict(synth.CLS_TINY)))
    cls.load_state_dict(tiny_classifier_sd())
    cls.attach(tiny)
    x = rnd((2, 4, 16, 64), 105)
    vf = synth.synthetic_cavp(2, 33, 64, seed=4321)
    grad, prob = tiny.engine.classifier_grad(x.cuda(), torch.tensor([500.0, 37.0]).cuda(), vf.cuda(), want_prob=True)
    assert torch.allclose(prob.cpu(), g["cls_p"], atol=2e-2)
    err = rel_l2(grad.cpu(), g["cls_grad"])
    print(f"tiny classifier grad rel-L2 {err:.3e}")
    assert err < 5e-2
    B = 2
    xT = synth.synthetic_xT(B, seed=21)
    c = tiny.get_learned_conditioning(synth.synthetic_cavp(B, 32, 64, seed=1234).cuda())
    uc = torch.zeros_like(c)
    for name, S in (("DDIM", 10), ("DPM_Solver", 10)):
        z, _ = tiny.sample_log_with_classifier_diff_sampler(
            c, origin_cond=vf.cuda(), batch_size=B, sampler_name=name, ddim_steps=S, unconditional_guidance_scale=4.5,
            unconditional_conditioning=uc, classifier=cls, classifier_guide_scale=50.0, x_T=xT.clone())
        err = rel_l2(z.cpu(), g[f"{name}_{S}_cg_z"])
        print(f"double guidance {name}-{S}: rel-L2 {err:.3e}")
        assert err < TRAJ_TOL


# ------------------------------------------------------------------------------------------- full config
def test_full_unet_forward_vs_golden(full):
    g = gold("g4_full_unet.npz")
    x, c = rnd((2, 4, 16, 64), 200), rnd((2, 32, 768), 201)
    y = full.apply_model(x.cuda(), torch.tensor([961, 41]).cuda(), c.cuda()).cpu()
    err = rel_l2(y, g["unet_y"])
    mae = (y - g["unet_y"]).abs().mean().item()
    print(f"full UNet forward: rel-L2 {err:.3e}  MAE {mae:.3e}")
    assert err < FWD_TOL
    y = full.apply_model(x.cuda(), torch.tensor([960.2, 40.96]).cuda(), c.cuda()).cpu()
    assert rel_l2(y, g["unet_y_float_t"]) < FWD_TOL


def test_full_vae_decode_vs_golden(full):
    g = gold("g4_full_unet.npz")
    d = full.decode_first_stage(rnd((1, 4, 16, 64), 202).cuda()).cpu()
    assert d.shape == (1, 3, 128, 512)
    err = rel_l2(d[:, 0], g["decode"])
    print(f"full VAE decode: rel-L2 {err:.3e}  MAE {(d[:, 0] - g['decode']).abs().mean().item():.3e}")
    assert err < FWD_TOL


def test_full_ddim_first4_steps_vs_golden(full):
    """Short trajectory (4 of 25 DDIM steps, CFG 4.5): close before chaotic divergence of the random-weight net."""
    from diff_foley_amd import synth
    g = gold("g5_full_samplers.npz")
    xT = synth.synthetic_xT(1, seed=21)
    c = full.get_learned_conditioning(synth.synthetic_cavp(1, 32, 512, seed=1234).cuda())
    assert rel_l2(c.cpu()[:, :2], g["cond_21"]) < 5e-3
    uc = torch.zeros_like(c)
    s = full._sampler("DDIM")
    s.make_schedule(25)
    from diff_foley_amd import engine as E
    full.engine.set_context(torch.cat([uc, c]))
    img = xT.cuda()
    steps = np.flip(s.ddim_timesteps)
    for i in range(4):
        idx = 25 - i - 1
        t = torch.full((1,), float(steps[i]), device="cuda")
        e = full.engine.unet_forward_cfg(img, t, 4.5)
        img, _ = E.ddim_update(img, e, s.ddim_alphas[idx], s.ddim_alphas_prev[idx], 0.0, s.ddim_sqrt_one_minus_alphas[idx])
    err = rel_l2(img.cpu(), g["ddim25_first4_x"])
    print(f"4 DDIM steps: rel-L2 {err:.3e}")
    assert err < TRAJ_TOL


def _bf16_ddim25_mel(full, seed):
    from diff_foley_amd import synth
    g = gold("g5_full_samplers.npz")
    xT = synth.synthetic_xT(1, seed=seed)
    c = full.get_learned_conditioning(synth.synthetic_cavp(1, 32, 512, seed=1234 + seed - 21).cuda())
    z, _ = full.sample_log_diff_sampler(c, 1, "DDIM", 25, unconditional_guidance_scale=4.5,
                                        unconditional_conditioning=torch.zeros_like(c), x_T=xT.clone())
    return full.decode_first_stage(z)[:, 0].cpu(), g[f"ddim25_mel_{seed}"]


@pytest.mark.xfail(strict=True, reason="bf16 MFMA operands (2^-9 per operand) put the 25-step mel at MAE ~4.5e-3: the north-star "
                                       "bound (< 1e-3, absolute) is met by the fp16-operand build only, which is the facade's "
                                       "default (tests/test_path_fp16_gpu.py::test_fp16_full_ddim25_mel_mae_absolute).  "
                                       "profiles/r6_bf16_error_budget.txt (tools/error_budget.py): bf16 WEIGHTS alone cost 2.0e-3 with "
                                       "every activation in fp16, so no per-op mix with bf16 weights can meet the bound")
def test_full_ddim25_mel_mae_north_star_bound_bf16(full):
    """The bound exactly as BASELINE.json states it -- mel-spec MAE < 1e-3 vs the CPU reference -- applied to the bf16 build.
    It does NOT hold (strict xfail: the day it does, this marker must go); what the bf16 build is held to is the
    range-normalised regression bound of test_full_ddim25_mel_mae_bf16_regression below."""
    mel, mr = _bf16_ddim25_mel(full, 21)
    assert (mel - mr).abs().mean().item() < 1e-3


def test_full_ddim25_mel_mae_bf16_regression(full):
    """Regression tripwire of the bf16-operand build (NOT the north-star tolerance, see the strict xfail above): decoded mel MAE
    vs the reference CPU sampler on identical seeds/inputs (B=1, 25-step DDIM, CFG 4.5; BASELINE.json config 1) below 1e-3 of
    the reference mel's range and below 1.5 % of its sigma."""
    from diff_foley_amd import synth
    g = gold("g5_full_samplers.npz")
    for seed in (21, 22):
        xT = synth.synthetic_xT(1, seed=seed)
        c = full.get_learned_conditioning(synth.synthetic_cavp(1, 32, 512, seed=1234 + seed - 21).cuda())
        uc = torch.zeros_like(c)
        z, _ = full.sample_log_diff_sampler(c, 1, "DDIM", 25, unconditional_guidance_scale=4.5,
                                            unconditional_conditioning=uc, x_T=xT.clone())
        mel = full.decode_first_stage(z)[:, 0].cpu()
        zr, mr = g[f"ddim25_z_{seed}"], g[f"ddim25_mel_{seed}"]
        mae = (mel - mr).abs().mean().item()
        span = (mr.max() - mr.min()).item()
        print(f"seed {seed}: z rel-L2 {rel_l2(z.cpu(), zr):.3e}; mel MAE {mae:.3e} (mel std {mr.std().item():.3f}, "
              f"range [{mr.min().item():.2f}, {mr.max().item():.2f}]); MAE on the [0,1]-normalised mel {mae / span:.3e}")
        # training mels live in [0,1] (data_preprocess/wav2spec.py:142-155); the random-weight decoder is not confined to it,
        # so the reference mel's own range is the unit here; the second bound keeps the bf16-vs-fp32 error below 1.5 % of sigma
        assert mae / span < 1e-3
        assert mae < 1.5e-2 * mr.std().item()


def test_full_dpm50_vs_golden(full):
    from diff_foley_amd import synth
    g = gold("g5_full_samplers.npz")
    xT = synth.synthetic_xT(1, seed=21)
    c = full.get_learned_conditioning(synth.synthetic_cavp(1, 32, 512, seed=1234).cuda())
    uc = torch.zeros_like(c)
    z, inter = full.sample_log_diff_sampler(c, 1, "DPM_Solver", 50, unconditional_guidance_scale=4.5,
                                            unconditional_conditioning=uc, x_T=xT.clone())
    assert inter is None
    mel = full.decode_first_stage(z)[:, 0].cpu()
    mr = g["dpm50_mel_21"]
    mae = (mel - mr).abs().mean().item()
    span = (mr.max() - mr.min()).item()
    print(f"DPM-50: z rel-L2 {rel_l2(z.cpu(), g['dpm50_z_21']):.3e}; mel MAE {mae:.3e}; normalised {mae / span:.3e}")
    assert mae / span < 1e-3 and mae < 1.5e-2 * mr.std().item()


def test_full_batch4_matches_batch1(full):
    """Config 2 shape (B=4 -> UNet batch 8): samples are independent, so row 0 of a B=4 run equals the B=1 run
    (size-independent property at BASELINE.json's full size)."""
    from diff_foley_amd import synth
    xT = synth.synthetic_xT(4, seed=21)
    c = full.get_learned_conditioning(synth.synthetic_cavp(4, 32, 512, seed=1234).cuda())
    uc = torch.zeros_like(c)
    z4, _ = full.sample_log_diff_sampler(c, 4, "DDIM", 5, unconditional_guidance_scale=4.5,
                                         unconditional_conditioning=uc, x_T=xT.clone())
    z1, _ = full.sample_log_diff_sampler(c[:1], 1, "DDIM", 5, unconditional_guidance_scale=4.5,
                                         unconditional_conditioning=uc[:1], x_T=xT[:1].clone())
    # different batch -> different tile/split-K choices -> different fp32 summation order -> different bf16
    # roundings downstream: the two runs agree to bf16 noise, not bit-exactly
    assert rel_l2(z4[:1].cpu(), z1.cpu()) < TRAJ_TOL


def test_full_classifier_forward_vs_golden(P, full):
    from diff_foley_amd import synth
    g = gold("g6_full_classifier.npz")
    cls = P.AlignmentClassifier(classifier_config=dict(params=dict(synth.CLS_FULL)))
    cls.load_state_dict(full_classifier_sd())
    cls.attach(full)
    x = rnd((2, 4, 16, 64), 205)
    vf = synth.synthetic_cavp(2, 33, 512, seed=4321)
    p = cls(x.cuda(), t=torch.tensor([500.0, 37.0]).cuda(), video_feat=vf.cuda()).cpu()      # keywords, as ddim.py:338 calls it
    assert torch.allclose(p, g["cls_p"], atol=2e-2), (p, g["cls_p"])
    grad = cls.log_prob_grad(x.cuda(), torch.tensor([500.0, 37.0]).cuda(), vf.cuda()).cpu()
    err = rel_l2(grad, g["cls_grad"])
    print(f"full classifier grad rel-L2 {err:.3e} (|grad| mean {g['cls_grad'].abs().mean().item():.3e})")
    assert err < 5e-2


def test_full_autotuned_plans_vs_golden(full):
    """The product (bench.py, the notebook path) runs AUTOTUNED plans: at full size the tuner reaches tile / split-K
    combinations no untuned plan uses (e.g. split-K on the SiLU-epilogue time-embedding GEMMs), so the goldens are checked
    through a freshly tuned engine as well."""
    g = gold("g4_full_unet.npz")
    full.autotune(True)
    full.engine.finalize()              # drop the untuned plans of the earlier tests
    try:
        x, c = rnd((2, 4, 16, 64), 200), rnd((2, 32, 768), 201)
        y = full.apply_model(x.cuda(), torch.tensor([961, 41]).cuda(), c.cuda()).cpu()
        err = rel_l2(y, g["unet_y"])
        print(f"autotuned full UNet forward: rel-L2 {err:.3e}")
        assert err < FWD_TOL
        d = full.decode_first_stage(rnd((1, 4, 16, 64), 202).cuda()).cpu()
        assert rel_l2(d[:, 0], g["decode"]) < FWD_TOL
    finally:
        full.autotune(False)
        full.engine.finalize()


def test_apply_model_context_cache_is_not_fooled_by_address_reuse(full):
    """apply_model reuses the hoisted context work only for the very same tensor object and version.  A NEW conditioning
    tensor that the caching allocator places at the address of a freed one must be picked up (round 2: a data_ptr()-keyed
    cache silently kept the old context)."""
    x, t = rnd((2, 4, 16, 64), 400).cuda(), torch.tensor([500, 37]).cuda()
    c1 = rnd((2, 32, 768), 401).cuda()
    y1 = full.apply_model(x, t, c1).cpu()
    addr = c1.data_ptr()
    del c1
    c2 = rnd((2, 32, 768), 402).cuda()           # same size: the allocator hands back the block just freed
    reused = c2.data_ptr() == addr
    y2 = full.apply_model(x, t, c2).cpu()
    c2b = c2.clone()
    y2_ref = full.apply_model(x, t, c2b).cpu()   # a different object: always re-set
    print(f"address reused: {reused}; |y2 - y1| rel {rel_l2(y2, y1):.3e}, |y2 - y2_ref| rel {rel_l2(y2, y2_ref):.3e}")
    assert rel_l2(y2, y2_ref) < 1e-6 and rel_l2(y2, y1) > 1e-2
    c2.mul_(0.5)                                 # in-place edit bumps the version counter: must be picked up as well
    y3 = full.apply_model(x, t, c2).cpu()
    assert rel_l2(y3, y2) > 1e-3


@pytest.mark.parametrize("Tc", [32, 17, 1])
def test_full_cross_attention_forms_agree(full, monkeypatch, Tc):
    """Cross-attention against operands precomputed from the context (engine.hip context_px: score GEMM with softmax
    epilogue + output GEMM) vs the q-projection -> attention kernel -> out-projection form (DF_NO_XPRE=1), also for context
    lengths below the 32-column head group (masked softmax).  attention_openai.py:152-194."""
    x, c = rnd((2, 4, 16, 64), 300 + Tc), rnd((2, Tc, 768), 301 + Tc)
    t = torch.tensor([700, 13]).cuda()
    full.engine.finalize()
    y_pre = full.apply_model(x.cuda(), t, c.cuda()).cpu()
    monkeypatch.setenv("DF_NO_XPRE", "1")
    full.engine.finalize()
    try:
        y_ref = full.apply_model(x.cuda(), t, c.cuda().clone()).cpu()
    finally:
        monkeypatch.delenv("DF_NO_XPRE")
        full.engine.finalize()
    err = rel_l2(y_pre, y_ref)
    print(f"Tc={Tc}: precomputed-context vs attention-kernel form rel-L2 {err:.3e}")
    assert torch.isfinite(y_pre).all() and err < 2e-2      # two bf16 roundings of the same math, each ~1e-2 from fp32


def test_tiny_autotuned_plans_match_untuned(P):
    """The on-device autotuner (isolated ranking + in-situ refinement, engine.hip autotune_plan) only changes tile /
    split-K choices: an autotuned engine must reproduce the golden vectors for every plan type."""
    from diff_foley_amd import synth
    m = P.LatentDiffusion(**P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY))
    m.load_state_dict(tiny_state_dict())
    m.cuda()
    m.autotune(True)
    g = gold("g3_tiny_unet.npz")
    x, c = rnd((2, 4, 16, 64), 102), rnd((2, 32, 128), 101)
    y = m.apply_model(x.cuda(), torch.tensor([500, 37]).cuda(), c.cuda()).cpu()
    assert rel_l2(y, g["y_int"]) < FWD_TOL
    d = m.decode_first_stage(rnd((2, 4, 16, 64), 103).cuda()).cpu()
    assert rel_l2(d, g["decode"]) < FWD_TOL
    g5 = gold("g5_tiny_samplers.npz")
    xT = synth.synthetic_xT(2, seed=21)
    cc = m.get_learned_conditioning(synth.synthetic_cavp(2, 32, 64, seed=1234).cuda())
    z, _ = m.sample_log_diff_sampler(cc, 2, "DDIM", 25, unconditional_guidance_scale=4.5,
                                     unconditional_conditioning=torch.zeros_like(cc), x_T=xT.clone())
    assert rel_l2(z.cpu(), g5["DDIM_25_z"]) < TRAJ_TOL
    g6 = gold("g6_tiny_classifier.npz")
    cls = P.AlignmentClassifier(classifier_config=dict(params=dict(synth.CLS_TINY)))
    cls.load_state_dict(tiny_classifier_sd())
    cls.attach(m)
    vf = synth.synthetic_cavp(2, 33, 64, seed=4321)
    grad, prob = m.engine.classifier_grad(rnd((2, 4, 16, 64), 105).cuda(), torch.tensor([500.0, 37.0]).cuda(), vf.cuda(),
                                          want_prob=True)
    assert torch.allclose(prob.cpu(), g6["cls_p"], atol=2e-2)
    assert rel_l2(grad.cpu(), g6["cls_grad"]) < 5e-2


def test_tune_cache_round_trip(P, tmp_path, monkeypatch):
    """DF_TUNE_CACHE: the first engine tunes and saves its (tile, split-K, walk) choices, a second one configures its
    plans from the file without trial launches and produces bit-identical results."""
    from diff_foley_amd import synth
    cache = tmp_path / "tune.txt"
    monkeypatch.setenv("DF_TUNE_CACHE", str(cache))
    x, c = rnd((2, 4, 16, 64), 102), rnd((2, 32, 128), 101)
    t = torch.tensor([500, 37])
    outs = []
    for _ in range(2):
        m = P.LatentDiffusion(**P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY))
        m.load_state_dict(tiny_state_dict())
        m.cuda()
        m.autotune(True)
        outs.append(m.apply_model(x.cuda(), t.cuda(), c.cuda()).cpu())
        assert cache.exists() and len(cache.read_text().splitlines()) > 5
        del m
    assert torch.equal(outs[0], outs[1])
    assert rel_l2(outs[0], gold("g3_tiny_unet.npz")["y_int"]) < FWD_TOL


def test_tiny_ragged_context_lengths_and_latent_sizes(tiny):
    """Edge shapes the reference API admits: context of 1 / 17 / 40 frames (pos_emb holds 40, video_feat_encoder.py:10),
    latent widths other than 64 (``size_len`` of sample_log_diff_sampler, ddpm.py:1287-1314) and the error path for a
    context longer than pos_emb."""
    usd, vsd, csd, synth = _oracle_tiny()
    from oracle import unet as ou
    for T in (1, 17, 40):
        feats = rnd((2, T, 64), 500 + T)
        c = tiny.get_learned_conditioning(feats.cuda())
        assert c.shape == (2, T, 128)
        x = rnd((2, 4, 16, 64), 600 + T)
        t = torch.tensor([999, 1])
        ref = ou.unet_forward(usd, synth.UNET_TINY, x, t, c.cpu())
        y = tiny.apply_model(x.cuda(), t.cuda(), c).cpu()
        assert rel_l2(y, ref) < FWD_TOL, T
    with pytest.raises(RuntimeError):
        tiny.get_learned_conditioning(rnd((1, 41, 64), 1).cuda())
    for (H, W) in ((16, 32), (16, 128), (8, 8)):
        x, c = rnd((1, 4, H, W), 700 + W), rnd((1, 32, 128), 701)
        t = torch.tensor([500])
        ref = ou.unet_forward(usd, synth.UNET_TINY, x, t, c)
        y = tiny.apply_model(x.cuda(), t.cuda(), c.cuda()).cpu()
        assert y.shape == ref.shape and rel_l2(y, ref) < FWD_TOL, (H, W)
    c = tiny.get_learned_conditioning(synth.synthetic_cavp(1, 32, 64, seed=3).cuda())
    z, _ = tiny.sample_log_diff_sampler(c, 1, "DDIM", 5, size_len=32, unconditional_guidance_scale=4.5,
                                        unconditional_conditioning=torch.zeros_like(c))
    assert z.shape == (1, 4, 16, 32) and torch.isfinite(z).all()


def test_tiny_decode_large_batch_is_chunked(tiny):
    """decode_first_stage on more than 16 samples runs in chunks of 16 with identical per-sample results."""
    z = rnd((19, 4, 16, 64), 808)
    full = tiny.decode_first_stage(z.cuda()).cpu()
    assert full.shape[0] == 19
    part = tiny.decode_first_stage(z[16:].cuda()).cpu()
    assert rel_l2(full[16:], part) < 2e-2          # different batch -> different plan/tiles: operand-rounding noise only
    one = tiny.decode_first_stage(z[:16].cuda()).cpu()
    assert torch.equal(full[:16], one)


# ------------------------------------------------------------------------------------------- notebook drop-in
def test_notebook_cells_replayed_verbatim(P):
    """inference/diff_foley_inference.ipynb cells 5-13 with only the import line changed:
    ``load_model_from_config`` = instantiate_from_config(config.model) -> load_state_dict(strict=False) -> .cuda() ->
    .eval() for the LDM *and* the double-guidance classifier, which is then handed STRAIGHT to
    sample_log_with_classifier_diff_sampler (no attach call exists in the notebook), the 32-frame window loop,
    decode_first_stage and the channel-0 slice.  Reference targets are used as written in the YAML files
    (Stage2_LDM.yaml:2, Double_Guidance_Classifier.yaml:3); sizes are the tiny configuration (numerics of this very
    sampler are pinned against the reference in test_tiny_classifier_grad_and_double_guidance_vs_golden)."""
    from diff_foley_amd import synth
    from diff_foley_amd.ldm import instantiate_from_config

    class Cfg(dict):                       # stands in for OmegaConf: attribute access on the loaded YAML
        __getattr__ = dict.__getitem__

    def load_model_from_config(config, sd, verbose=False):          # cell 5, torch.load replaced by the dict itself
        model = instantiate_from_config(config.model)
        m, u = model.load_state_dict(sd, strict=False)
        model.cuda()
        model.eval()
        return model

    ldm_config = Cfg(model=dict(target="diff_foley.models.diffusion.ddpm.LatentDiffusion",
                                params=P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY)))
    latent_diffusion_model = load_model_from_config(ldm_config, tiny_state_dict())                      # cell 6
    classifier_config = Cfg(model=dict(
        target="diff_foley.modules.double_guidance.alignment_classifier.Alignment_Classifier_Double_Guidance",
        params=dict(linear_start=0.00085, linear_end=0.0120, timesteps=1000, scale_factor=0.18215, first_stage_key="spec",
                    classifier_config=dict(target="diff_foley.modules.double_guidance.alignment_backbone.Classifier_Backbone",
                                           params=dict(image_size=32, use_spatial_transformer=True, transformer_depth=1,
                                                       use_checkpoint=True, legacy=False, **synth.CLS_TINY)))))
    classifier = load_model_from_config(classifier_config, tiny_classifier_sd())                        # cell 12
    device = torch.device("cuda")
    sample_num, cfg_scale, cg_scale, steps, sampler = 2, 4.5, 50, 10, "DPM_Solver"                        # cell 13
    cavp_feats = synth.synthetic_cavp(1, 33, 64, seed=4321)[0].numpy()         # (33, 64): what extract_cavp returns
    video_feat = torch.from_numpy(cavp_feats).unsqueeze(0).repeat(sample_num, 1, 1).to(device)
    feat_len = video_feat.shape[1]
    truncate_len = 32
    window_num = feat_len // truncate_len
    mels = []
    for i in range(window_num):
        start, end = i * truncate_len, (i + 1) * truncate_len
        embed_cond_feat = latent_diffusion_model.get_learned_conditioning(video_feat[:, start:end])
        uncond_cond = torch.zeros(embed_cond_feat.shape).to(device)
        audio_samples, _ = latent_diffusion_model.sample_log_with_classifier_diff_sampler(
            embed_cond_feat, origin_cond=video_feat, batch_size=video_feat.shape[0], sampler_name=sampler, ddim_steps=steps,
            unconditional_guidance_scale=cfg_scale, unconditional_conditioning=uncond_cond, classifier=classifier,
            classifier_guide_scale=cg_scale, x_T=synth.synthetic_xT(sample_num, seed=21).to(device))
        assert audio_samples.shape == (sample_num, 4, 16, 64)
        audio_samples = latent_diffusion_model.decode_first_stage(audio_samples)
        audio_samples = audio_samples[:, 0, :, :].detach().cpu().numpy()
        mels.append(audio_samples)
    assert window_num == 1 and mels[0].shape[0] == sample_num and np.isfinite(mels[0]).all()
    # second, keyword-free call: the classifier protocol of the reference (x_in, t=t, video_feat=c) also works directly
    p = classifier(torch.zeros(sample_num, 4, 16, 64, device=device), t=torch.full((sample_num,), 500.0, device=device),
                   video_feat=video_feat)
    assert p.shape == (sample_num, 1) and bool(((p > 0) & (p < 1)).all())
    # unsupported reference kwargs raise instead of being dropped
    with pytest.raises(NotImplementedError):
        latent_diffusion_model.sample_log_diff_sampler(embed_cond_feat, batch_size=sample_num, sampler_name="DDIM",
                                                       ddim_steps=5, quantize_x0=True)
    with pytest.raises(ValueError, match="mask"):         # inpainting is on the path; a mask that is no [B][1|C][H][W] map is refused
        latent_diffusion_model.sample_log_diff_sampler(embed_cond_feat, batch_size=sample_num, sampler_name="DDIM",
                                                       ddim_steps=5, mask=torch.ones(1), x0=torch.zeros(1))
    with pytest.raises(AssertionError):
        latent_diffusion_model.sample_log_diff_sampler(embed_cond_feat, batch_size=sample_num, sampler_name="DPM_Solver",
                                                       ddim_steps=1)
    with pytest.raises(RuntimeError, match="expects"):
        latent_diffusion_model.get_learned_conditioning(torch.zeros(1, 32, 512, device=device))     # raw dim of the FULL model


def test_reloading_weights_invalidates_packed_copies(P):
    """a second load_state_dict on a used model must not keep running on the old packed operands."""
    from diff_foley_amd import synth
    x, c = rnd((2, 4, 16, 64), 102), rnd((2, 32, 128), 101)
    t = torch.tensor([500, 37])
    spec = synth.state_dict_spec(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY)
    sd0, sd1 = tiny_state_dict(0), synth.make_state_dict(spec, 1)
    m = P.LatentDiffusion(**P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY))
    m.load_state_dict(sd0)
    m.cuda()
    y0 = m.apply_model(x.cuda(), t.cuda(), c.cuda()).cpu()
    m.load_state_dict(sd1)
    y1 = m.apply_model(x.cuda(), t.cuda(), c.cuda()).cpu()
    fresh = P.LatentDiffusion(**P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY))
    fresh.load_state_dict(sd1)
    fresh.cuda()
    y1_ref = fresh.apply_model(x.cuda(), t.cuda(), c.cuda()).cpu()
    assert rel_l2(y0, y1_ref) > 0.1                     # the two checkpoints really differ
    assert torch.equal(y1, y1_ref)
    # in-place mutation of the conditioning tensor is seen (apply_model keys its K/V cache on the tensor version)
    cc = c.cuda().clone()
    ya = m.apply_model(x.cuda(), t.cuda(), cc).cpu()
    cc.zero_()
    yb = m.apply_model(x.cuda(), t.cuda(), cc).cpu()
    assert rel_l2(ya, yb) > 1e-3


def test_plan_cache_is_bounded(P, monkeypatch):
    """plans are cached per shape and own their workspaces; many shapes must not grow HBM without bound."""
    from diff_foley_amd import synth
    monkeypatch.setenv("DF_MAX_PLANS", "6")          # read once per process: effective only if no plan was evicted before
    m = P.LatentDiffusion(**P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY))
    m.load_state_dict(tiny_state_dict())
    m.cuda()
    c = rnd((1, 32, 128), 1).cuda()
    sizes = []
    for w in (16, 32, 48, 64, 80, 96, 112, 128, 144, 160, 176, 192, 208, 224, 240, 256, 16, 32) * 3:
        x = rnd((1, 4, 16, w), w).cuda()
        y = m.apply_model(x, torch.tensor([10.0]).cuda(), c)
        assert torch.isfinite(y).all()
        sizes.append(m.engine.plan_count())
    nmax = max(n for n, _ in sizes)
    assert nmax <= 32, sizes[-1]                      # the default bound (or the smaller one set above)
    assert sizes[-1][1] < 8 * (1 << 30)


def test_debug_checksums_and_tune_cache_round_trip(P):
    """Round-3 debug / distribution entry points: (i) the per-op workspace checksums of two identical runs are identical and
    labelled op by op (df_debug_checksums: the tool that localised the ring race); (ii) the autotuner's choices exported as text
    and imported again reproduce themselves (df_tune_cache_export / _import: what broadcast_packed_model ships to the ranks)."""
    from diff_foley_amd import synth
    m = P.LatentDiffusion(**P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY))
    m.load_state_dict(synth.make_state_dict(synth.state_dict_spec(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY), 0))
    m.cuda()
    m.autotune(True)
    feats = synth.synthetic_cavp(2, 32, 64, seed=1234).cuda()
    xT = synth.synthetic_xT(2).cuda()

    def run():
        c = m.get_learned_conditioning(feats)
        z, _ = m.sample_log_diff_sampler(c, 2, "DDIM", 3 + 1, unconditional_guidance_scale=4.5,
                                         unconditional_conditioning=torch.zeros_like(c), x_T=xT)
        return z

    z0 = run()                                          # builds and tunes the plans
    eng = m.engine
    seqs = []
    for _ in range(2):
        eng.debug_checksums(True, 1 << 14)
        z = run()
        seqs.append(eng.debug_checksums_read())
        assert torch.equal(z, z0)
    assert len(seqs[0]) > 100 and seqs[0] == seqs[1]
    lab = eng.debug_checksum_label(len(seqs[0]) - 1)     # labels live until the next debug_checksums() call
    assert "#" in lab and ":" in lab
    eng.debug_checksums(False)
    text = eng.tune_cache_export()
    assert text.count(b"\n") >= 10
    eng.tune_cache_import(text)                         # idempotent
    assert eng.tune_cache_export() == text
    with pytest.raises(RuntimeError):
        eng.tune_cache_import(b"bogus_key 999 1 0\n")   # tile id out of range


def test_full_cost_model_plan_is_reproducible_run_to_run():
    """Regression test of the round-6 root cause (csrc/common.h df_entry_touch): the FULL bf16 model on the cost-model plans
    (DF_TUNED_DEFAULTS=0 -- the plan whose st.ffproj kernel exposed it), with the short GELU epilogue (v_rcp_f32 erf, the build
    that failed in round 5), repeated CFG forwards under per-op workspace checksums: every repetition must leave the very bytes
    of the first one after every op, and every output must be finite.  (tools/race_hunt.py in a child process: the shipped plan
    table is imported once per process, so the cost-model plans need their own.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DF_TUNED_DEFAULTS="0", HUNT_PREC="bf16")
    env.pop("DF_LIB_OVERRIDE", None)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "race_hunt.py"), "6"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["diverged"] == 0 and d["nonfinite_outputs"] == 0, d


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_shipped_plan_is_reproducible_run_to_run(prec):
    """The plan the product runs (the shipped table): repeated CFG forwards of the full model leave the same bytes after every op.
    Round 6: the regenerated table put the fused QKV projection of the 320-channel level on a tile with two blocks per CU, and its
    V^T rows lost their bias at random -- a packed add with the operand select on src1 next to another block's MFMAs
    (csrc/common.h pk_add_hi, tools/check_pk_opsel.py); 7 of 12 repetitions differed."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HUNT_PREC=prec)
    for k in ("DF_LIB_OVERRIDE", "DF_TUNED_DEFAULTS", "DF_TUNED_TABLE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "race_hunt.py"), "10"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["diverged"] == 0 and d["nonfinite_outputs"] == 0, d


def test_facade_runs_the_shipped_plan_table(P):
    """The drop-in never calls a tuning step (inference/diff_foley_inference.ipynb:80-95 has none): LatentDiffusion(...).cuda()
    imports the shipped table of this GPU model (diff_foley_amd/tuned/, engine.load_tuned_defaults) and every GEMM of the
    BASELINE configs[1] step then runs the (tile, split-K) the table holds for its shape -- the kernels bench.py times."""
    import csv
    import os
    from diff_foley_amd import synth
    m = P.LatentDiffusion(**P.stage2_config())          # default precision, no autotune call anywhere
    m.load_state_dict(full_state_dict())
    m.cuda()
    eng = m.engine
    if os.environ.get("DF_TUNED_DEFAULTS", "1") == "0":
        pytest.skip("DF_TUNED_DEFAULTS=0")
    assert eng.tuned_defaults and os.path.exists(eng.tuned_defaults), "no shipped table for this device"
    table = {}
    for line in open(eng.tuned_defaults):
        key, tile, sk, gm = line.split()
        f = key.split("_")
        table.setdefault(tuple(int(v) for v in f[:4]), set()).add((int(tile), int(sk)))
    B = 4
    c = m.get_learned_conditioning(synth.synthetic_cavp(B).cuda())
    xT = synth.synthetic_xT(B).cuda()
    m.sample_log_diff_sampler(c, B, "DDIM", 2, unconditional_guidance_scale=4.5, unconditional_conditioning=torch.zeros_like(c), x_T=xT)
    eng.profile_begin()
    z, _ = m.sample_log_diff_sampler(c, B, "DDIM", 2, unconditional_guidance_scale=4.5, unconditional_conditioning=torch.zeros_like(c), x_T=xT)
    eng.profile_end()
    path = "/tmp/df_shipped_plan_ops.csv"
    eng.profile_dump(path)
    rows = [r for r in csv.DictReader(open(path)) if int(r["K"]) > 0 and r["tag"] != "out.conv"]
    assert len(rows) > 200
    missing = [r for r in rows if (int(r["M"]), int(r["N"]), int(r["K"]), int(r["taps"])) not in table]
    assert not missing, missing[:3]
    off = [r for r in rows if (int(r["tile"]), int(r["splitk"])) not in table[(int(r["M"]), int(r["N"]), int(r["K"]), int(r["taps"]))]]
    assert not off, off[:3]
    assert torch.isfinite(z).all()
    # the tuned plan uses the producer-specialised / persistent tiles; the cost model never picks them
    assert any(int(r["tile"]) >= 18 for r in rows)


def test_batch_outside_the_shipped_table_takes_its_nearest_entries(full):
    """A sampler batch the shipped table does not hold (B = 10: UNet batch 20) builds its plan from the entries of the nearest row
    count (engine.hip nearest_tune_choice: another batch size changes M only), not from the cost model -- the plan carries the
    producer-specialised / wide tiles -- and, samples being independent, its row 0 equals the B = 1 run to operand-rounding noise."""
    import csv
    import os
    from diff_foley_amd import synth
    if os.environ.get("DF_TUNED_DEFAULTS", "1") == "0" or not full.engine.tuned_defaults:
        pytest.skip("no shipped table in this process")
    B = 10
    xT = synth.synthetic_xT(B, seed=21)
    c = full.get_learned_conditioning(synth.synthetic_cavp(B, 32, 512, seed=1234).cuda())
    uc = torch.zeros_like(c)
    eng = full.engine
    zB, _ = full.sample_log_diff_sampler(c, B, "DDIM", 4, unconditional_guidance_scale=4.5, unconditional_conditioning=uc, x_T=xT.clone())
    eng.profile_begin()
    full.sample_log_diff_sampler(c, B, "DDIM", 1 + 1, unconditional_guidance_scale=4.5, unconditional_conditioning=uc, x_T=xT.clone())
    eng.profile_end()
    eng.profile_dump("/tmp/df_b10_ops.csv")
    rows = [r for r in csv.DictReader(open("/tmp/df_b10_ops.csv")) if int(r["K"]) > 0]
    table_m = {int(line.split("_")[0]) for line in open(eng.tuned_defaults)}
    assert any(int(r["M"]) not in table_m for r in rows)                      # the shape class really is outside the table
    ff1 = [r for r in rows if r["tag"] == "st.ff1"]
    assert ff1 and all(int(r["tile"]) >= 18 for r in ff1), [(r["M"], r["tile"]) for r in ff1][:6]
    assert sum(int(r["tile"]) >= 18 for r in rows) > len(rows) // 3
    z1, _ = full.sample_log_diff_sampler(c[:1], 1, "DDIM", 4, unconditional_guidance_scale=4.5, unconditional_conditioning=uc[:1],
                                         x_T=xT[:1].clone())
    assert torch.isfinite(zB).all() and rel_l2(zB[:1].cpu(), z1.cpu()) < TRAJ_TOL
