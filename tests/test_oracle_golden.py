"""CPU: the oracle (oracle/*.py) against every golden vector produced by the reference
(tests/golden/make_golden.py).  fp32 on both sides, so tolerances are tight (the two
differ only in op association order)."""
import numpy as np
import pytest
import torch

from helpers import gold, rnd, tiny_state_dict, tiny_classifier_sd, full_state_dict, full_classifier_sd
from diff_foley_amd import synth
from oracle import schedule as osch, unet as ou, vae as ov, samplers as osamp

TOL = 2e-5


def close(a, b, tol=TOL):
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a.double() - b.double()).abs().max().item()
    assert err <= tol * max(1.0, b.abs().max().item()), err


def test_g1_ddpm_buffers():
    g = gold("g1_schedules.npz")
    s = osch.ddpm_schedule()
    for k, v in s.items():
        assert torch.equal(v, g[k]), k


@pytest.mark.parametrize("S", [25, 50])
def test_g1_ddim_tables(S):
    g = gold("g1_schedules.npz")
    sch = osch.ddim_schedule(osch.ddpm_schedule()["alphas_cumprod"], S)
    assert np.array_equal(sch["timesteps"], g[f"ddim{S}_timesteps"].numpy())
    assert np.array_equal(np.asarray(sch["alphas"], dtype=np.float64), g[f"ddim{S}_alphas"].numpy())
    assert np.array_equal(np.asarray(sch["alphas_prev"], dtype=np.float64), g[f"ddim{S}_alphas_prev"].numpy())
    assert np.array_equal(np.asarray(sch["sqrt_one_minus_alphas"], dtype=np.float64),
                          g[f"ddim{S}_sqrt_one_minus_alphas"].numpy())
    assert np.all(np.asarray(sch["sigmas"], dtype=np.float64) == 0)


@pytest.mark.parametrize("S", [10, 25, 50])
@pytest.mark.parametrize("eta", [0, 1])
def test_g11_ddim_quad_tables(S, eta):
    """ddim_discretize="quad" (util.py:50-51) against the reference's own make_schedule output."""
    g = gold("g11_ddim_quad.npz")
    sch = osch.ddim_schedule(osch.ddpm_schedule()["alphas_cumprod"], S, eta=float(eta), discretize="quad")
    tag = f"quad{S}_eta{eta}"
    assert np.array_equal(sch["timesteps"], g[f"{tag}_timesteps"].numpy())
    assert np.array_equal(np.asarray(sch["alphas"], dtype=np.float64), g[f"{tag}_alphas"].numpy())
    assert np.array_equal(np.asarray(sch["alphas_prev"], dtype=np.float64), g[f"{tag}_alphas_prev"].numpy())
    assert np.array_equal(np.asarray(sch["sqrt_one_minus_alphas"], dtype=np.float64), g[f"{tag}_sqrt_one_minus_alphas"].numpy())
    assert np.allclose(np.asarray(sch["sigmas"], dtype=np.float64), g[f"{tag}_sigmas"].numpy(), rtol=1e-6, atol=0)


@pytest.mark.parametrize("S", [25, 50])
def test_g1_dpm_tables(S):
    g = gold("g1_schedules.npz")
    ns = osch.NoiseScheduleVP(osch.ddpm_schedule()["alphas_cumprod"])
    t = torch.linspace(1.0, 1.0 / 1000, S + 1)
    assert torch.equal(t, g[f"dpm{S}_t"])
    close(ns.marginal_lambda(t), g[f"dpm{S}_lambda"], 1e-6)
    close(ns.marginal_alpha(t), g[f"dpm{S}_alpha"], 1e-6)
    close(ns.marginal_std(t), g[f"dpm{S}_sigma"], 1e-6)
    close(ns.marginal_log_mean_coeff(g["interp_t"]), g["interp_log_alpha"], 1e-6)


def test_g2_timestep_embedding():
    g = gold("g1_schedules.npz")
    assert torch.equal(ou.timestep_embedding(g["temb_t"], 320), g["temb_320"])
    assert torch.equal(ou.timestep_embedding(g["temb_t"], 64), g["temb_64"])


def _tiny_parts():
    sd = tiny_state_dict()
    return (ou.sub_state_dict(sd, "model.diffusion_model."), ou.sub_state_dict(sd, "first_stage_model."),
            ou.sub_state_dict(sd, "cond_stage_model."))


def test_g3_tiny_ops():
    g = gold("g3_tiny_ops.npz")
    usd, _, _ = _tiny_parts()
    cfg = synth.UNET_TINY
    emb = ou.time_embed(usd, cfg, g["t"])
    c = g["c"]
    h = cfg["num_heads"]
    close(ou.unet_forward(usd, cfg, g["x"], g["t"], c), g["y"])
    for p in ("input_blocks.1.0", "input_blocks.4.0", "output_blocks.5.0"):
        close(g[p + "__in1"], emb)
        close(ou.resblock(usd, p, g[p + "__in"], emb), g[p + "__out"])
    for p in ("input_blocks.1.1", "middle_block.1", "output_blocks.5.1"):
        close(ou.spatial_transformer(usd, p, g[p + "__in"], c, h), g[p + "__out"])
    close(ou._run_block(usd, [("down", "input_blocks.3.0")], g["input_blocks.3.0__in"], emb, c, h),
          g["input_blocks.3.0__out"])
    close(ou._run_block(usd, [("up", "output_blocks.2.1")], g["output_blocks.2.1__in"], emb, c, h),
          g["output_blocks.2.1__out"])


def test_g3_tiny_unet_vae_cond():
    g = gold("g3_tiny_unet.npz")
    usd, vsd, csd = _tiny_parts()
    x, c = rnd((2, 4, 16, 64), 102), rnd((2, 32, 128), 101)
    close(ou.unet_forward(usd, synth.UNET_TINY, x, torch.tensor([500, 37]), c), g["y_int"])
    close(ou.unet_forward(usd, synth.UNET_TINY, x, torch.tensor([500.25, 37.7]), c), g["y_flt"])
    close(ov.decode_first_stage(vsd, synth.VAE_TINY, rnd((2, 4, 16, 64), 103)), g["decode"])
    close(ov.cond_stage(csd, rnd((2, 32, 64), 104)), g["cond"])


def _tiny_sampling_setup():
    usd, vsd, csd = _tiny_parts()
    B = 2
    xT = synth.synthetic_xT(B, seed=21)
    c = ov.cond_stage(csd, synth.synthetic_cavp(B, 32, 64, seed=1234))
    uc = torch.zeros_like(c)
    apply_model = lambda x, t, cc: ou.unet_forward(usd, synth.UNET_TINY, x, t, cc)
    return apply_model, xT, c, uc, vsd


def test_g5_tiny_samplers():
    g = gold("g5_tiny_samplers.npz")
    apply_model, xT, c, uc, vsd = _tiny_sampling_setup()
    acp = osch.ddpm_schedule()["alphas_cumprod"]
    tol = 2e-4      # 25-50 fp32 steps of a random-weight UNet: association-order noise accumulates
    for S in (25, 50):
        z, inter = osamp.ddim_sample(apply_model, acp, S, xT, c, 4.5, uc)
        close(z, g[f"DDIM_{S}_z"], tol)
        close(inter["pred_x0"][-1], g[f"DDIM_{S}_pred_x0_last"], tol)
        assert len(inter["x_inter"]) == int(g[f"DDIM_{S}_n_inter"])
        if S == 25:
            close(ov.decode_first_stage(vsd, synth.VAE_TINY, z), g["DDIM_25_mel"], tol)
    z, _ = osamp.ddim_sample(apply_model, acp, 25, xT, c)
    close(z, g["DDIM_25_nocfg_z"], tol)
    for S in (25, 10):
        z, _ = osamp.dpm_solver_sample(apply_model, acp, S, xT, c, 4.5, uc)
        close(z, g[f"DPM_Solver_{S}_z"], tol)
    z, inter = osamp.plms_sample(apply_model, acp, 25, xT, c, 4.5, uc)
    close(z, g["PLMS_25_z"], tol)
    close(inter["pred_x0"][-1], g["PLMS_25_pred_x0_last"], tol)
    torch.manual_seed(77)
    z, _ = osamp.ddpm_sample(apply_model, osch.ddpm_schedule(), xT, c, timesteps=6)
    close(z, g["ancestral_6_z"], tol)


def test_g10_tiny_inpainting():
    """mask / x0 through the oracle's DDIM, PLMS and ancestral loops against the reference's own inpainting runs (G10): the
    known region is re-noised with q_sample(x0, t) every step and pasted over the sample (ddim.py:206-209, plms.py:147-150,
    ddpm.py:1239-1241).  q_sample's noise is replayed from the generator the golden script fed it from."""
    g = gold("g10_tiny_inpaint.npz")
    apply_model, xT, c, uc, vsd = _tiny_sampling_setup()
    acp = osch.ddpm_schedule()["alphas_cumprod"]
    x0, mask = g["x0"], g["mask"]
    gq = torch.Generator()
    qn = lambda shape: torch.randn(tuple(shape), generator=gq)
    tol = 2e-4
    gq.manual_seed(int(g["q_seed"]))
    z, _ = osamp.ddim_sample(apply_model, acp, 6, xT, c, 4.5, uc, mask=mask, x0=x0, q_noise_fn=qn)
    close(z, g["DDIM_6_z"], tol)
    gq.manual_seed(int(g["q_seed"]))
    z, _ = osamp.plms_sample(apply_model, acp, 6, xT, c, 4.5, uc, mask=mask, x0=x0, q_noise_fn=qn)
    close(z, g["PLMS_6_z"], tol)
    gq.manual_seed(int(g["q_seed"]))
    torch.manual_seed(77)
    z, _ = osamp.ddpm_sample(apply_model, osch.ddpm_schedule(), xT, c, timesteps=4, mask=mask, x0=x0, q_noise_fn=qn)
    close(z, g["ancestral_4_z"], tol)
    # the masked-in region of the result is dominated by x0 only through the last blend: sanity of the mask's orientation
    assert float((z - xT).abs().mean()) > 0


def test_g12_tiny_stochastic_ddim_temperature_and_noise_dropout():
    """eta > 0 DDIM through the oracle against the reference's own runs (G12): sigma_t * noise * temperature, then
    torch.nn.functional.dropout on the noise (ddim.py:269-271).  The reference drew the noise and the dropout mask from the global
    CPU generator in that order; the oracle draws in the same order after the same seed."""
    g = gold("g12_tiny_ddim_stochastic.npz")
    apply_model, xT, c, uc, vsd = _tiny_sampling_setup()
    acp = osch.ddpm_schedule()["alphas_cumprod"]
    for tag, eta, kw in (("eta1", 1.0, dict()), ("eta1_temp07", 1.0, dict(temperature=0.7)),
                         ("eta1_drop025", 1.0, dict(noise_dropout=0.25)),
                         ("eta05_drop05_temp13", 0.5, dict(noise_dropout=0.5, temperature=1.3))):
        torch.manual_seed(int(g["noise_seed"]))
        z, _ = osamp.ddim_sample(apply_model, acp, 6, xT, c, 4.5, uc, eta=eta, **kw)
        close(z, g[f"DDIM_6_{tag}_z"], 2e-4)
    # the dropout really acts: with it the run differs from the plain eta = 1 run by far more than the tolerance
    assert float((g["DDIM_6_eta1_drop025_z"] - g["DDIM_6_eta1_z"]).abs().mean()) > 1e-2


def test_g6_tiny_classifier_and_double_guidance():
    g = gold("g6_tiny_classifier.npz")
    csd = ou.sub_state_dict(tiny_classifier_sd(), "model.")
    cls = lambda x, t, vf: ou.classifier_forward(csd, synth.CLS_TINY, x, t, vf)
    x = rnd((2, 4, 16, 64), 105)
    vf = synth.synthetic_cavp(2, 33, 64, seed=4321)
    tt = torch.tensor([500, 37])
    close(cls(x, tt, vf), g["cls_p"], 1e-5)
    close(osamp.classifier_grad(cls, x, tt, vf), g["cls_grad"], 1e-4)
    apply_model, xT, c, uc, _ = _tiny_sampling_setup()
    acp = osch.ddpm_schedule()["alphas_cumprod"]
    z, _ = osamp.ddim_sample(apply_model, acp, 10, xT, c, 4.5, uc, classifier=cls, origin_cond=vf,
                             classifier_scale=50.0)
    close(z, g["DDIM_10_cg_z"], 5e-4)
    z, _ = osamp.dpm_solver_sample(apply_model, acp, 10, xT, c, 4.5, uc, classifier=cls, origin_cond=vf,
                                   classifier_scale=50.0)
    close(z, g["DPM_Solver_10_cg_z"], 5e-4)


@pytest.mark.slow
def test_g4_full_unet_forward():
    g = gold("g4_full_unet.npz")
    sd = full_state_dict()
    usd = ou.sub_state_dict(sd, "model.diffusion_model.")
    x, c = rnd((2, 4, 16, 64), 200), rnd((2, 32, 768), 201)
    close(ou.unet_forward(usd, synth.UNET_FULL, x, torch.tensor([961, 41]), c), g["unet_y"], 5e-5)
    close(ou.unet_forward(usd, synth.UNET_FULL, x, torch.tensor([960.2, 40.96]), c), g["unet_y_float_t"], 5e-5)
    vsd = ou.sub_state_dict(sd, "first_stage_model.")
    close(ov.decode_first_stage(vsd, synth.VAE_FULL, rnd((1, 4, 16, 64), 202))[:, 0], g["decode"], 5e-5)


@pytest.mark.slow
def test_g6_full_classifier():
    g = gold("g6_full_classifier.npz")
    csd = ou.sub_state_dict(full_classifier_sd(), "model.")
    cls = lambda x, t, vf: ou.classifier_forward(csd, synth.CLS_FULL, x, t, vf)
    x = rnd((2, 4, 16, 64), 205)
    vf = synth.synthetic_cavp(2, 33, 512, seed=4321)
    tt = torch.tensor([500, 37])
    close(cls(x, tt, vf), g["cls_p"], 1e-5)
    close(osamp.classifier_grad(cls, x, tt, vf), g["cls_grad"], 1e-4)


def test_g7_cavp_oracle_matches_reference_vectors():
    """CAVP video encoder restatement (oracle/cavp.py) against vectors produced by the reference's own
    ResNet3dSlowOnly / CAVP_Inference code (mmcv.ConvModule stand-in declared in oracle/ref_import.py)."""
    from oracle import cavp as ocavp
    from diff_foley_amd import synth
    g = gold("g7_cavp_tiny.npz")
    sd = synth.make_state_dict(synth.cavp_spec(synth.CAVP_TINY))
    v = synth.synthetic_video(1, 4, 64, seed=77)
    f = ocavp.encode_video(sd, v, stage_blocks=tuple(synth.CAVP_TINY["stage_blocks"]))
    assert torch.allclose(f, g["feats"], atol=1e-5)
    f = ocavp.encode_video(sd, v, normalize=False, stage_blocks=tuple(synth.CAVP_TINY["stage_blocks"]))
    assert torch.allclose(f, g["feats_raw"], atol=1e-4, rtol=1e-4)
    g = gold("g7_cavp_full.npz")
    sd = synth.make_state_dict(synth.cavp_spec())
    v = synth.synthetic_video(1, 8, 224, seed=77)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    f = ocavp.encode_video(sd, v)
    assert torch.allclose(f, g["feats"], atol=1e-5)
