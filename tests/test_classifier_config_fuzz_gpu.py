"""Randomised differential test of the hand-written classifier backward pass (csrc/backward.hip, engine.hip:build_classifier_grad)
over classifier CONFIGURATIONS: seeded draws of the Classifier_Backbone's constructor arguments (alignment_backbone.py:420-640:
model_channels, channel_mult, num_res_blocks, attention_resolutions, num_heads, context_dim), procedurally generated weights, and
``d sum(log p) / d x`` (cal_classifier_loglikelihood_grad, ddim.py:333-341) from ``df_classifier_grad`` against torch autograd through
the oracle's fp32 forward (oracle/unet.py:classifier_forward) at a random latent size, batch and number of video frames.  The golden
G6 pins the tiny and the full configuration; the tape builder has a branch per block kind and per channel change, which this walks.

Tolerance (fp16-operand build): probability within 5e-3, gradient rel-L2 < 1.5e-2 (the full-size configuration measures 3.7e-3)."""
import os

import numpy as np
import pytest
import torch

from helpers import fuzz_seeds, rel_l2, tiny_state_dict

pytestmark = pytest.mark.gpu

N_CASES = 10


PREC = os.environ.get("DF_FUZZ_PREC", "fp16")          # exploratory: the bf16-operand build (8 x the rounding unit), in-plan autotuner
PREC_SCALE = 8.0 if PREC == "bf16" else 1.0
TUNE = os.environ.get("DF_FUZZ_TUNE", "0") != "0"
WIDE = os.environ.get("DF_FUZZ_WIDE", "0") != "0"      # exploratory sweeps: wider maps, batches and widths than the suite draws


def _draw(seed):
    r = np.random.default_rng(5200 + seed)
    while True:
        mc = int(r.choice([64, 128, 192, 320] if WIDE else [64, 128]))
        mult = [list(m) for m in ([1, 2], [1, 2, 2], [1, 1, 2], [1, 2, 4], [1, 1])][int(r.integers(0, 5))]
        nrb = int(r.choice([1, 2]))
        levels = len(mult)
        att = sorted(int(2 ** i) for i in range(levels) if r.random() < 0.6)
        chs = {mc * mult[i] for i in range(levels) if 2 ** i in att} | {mc * mult[-1]}
        heads = [h for h in (1, 2, 4, 8, 16) if all(ch % h == 0 and ch // h in (32, 64) for ch in chs)]      # the backward's head dims
        if heads:
            break
    cfg = dict(in_channels=4, out_channels=1, model_channels=mc, attention_resolutions=att, num_res_blocks=nrb, channel_mult=mult,
               num_heads=int(r.choice(heads)), context_dim=int(r.choice([64, 128, 512])))
    q = 2 ** (levels - 1)
    if WIDE:
        H = int(r.choice([h for h in (8, 16, 24, 32) if h % q == 0]))
        W = int(r.choice([w for w in (8, 16, 24, 32, 40, 64, 96) if w % q == 0]))
        return cfg, dict(B=int(r.choice([1, 2, 3, 4, 5, 8])), H=H, W=W, T=int(r.choice([1, 8, 31, 32, 33, 40])))
    H = int(r.choice([h for h in (8, 16) if h % q == 0]))
    W = int(r.choice([w for w in (16, 32, 64) if w % q == 0]))
    return cfg, dict(B=int(r.choice([1, 2, 3])), H=H, W=W, T=int(r.choice([8, 32, 33])))


# seeds a wider sweep (DF_FUZZ_SEED0=100 DF_FUZZ_CASES=60, round 6) failed on: model_channels 64 with channel_mult [1, 1] -- the head's
# conv halves 64 channels to 32, and the backward-data packing refused a Cout that is not a multiple of the 64-channel K step
REGRESSION_SEEDS = [119, 152]


@pytest.mark.parametrize("seed", sorted(set(fuzz_seeds(N_CASES)) | set(REGRESSION_SEEDS)))
def test_classifier_gradient_product_vs_autograd(seed):
    import diff_foley_amd as P
    from diff_foley_amd import synth
    from oracle import unet as ou, samplers as osamp
    cfg, o = _draw(seed)
    host = P.LatentDiffusion(precision=PREC, **P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY))
    host.load_state_dict(tiny_state_dict())
    host.cuda()
    if TUNE:
        host.autotune(True)
    sd = synth.make_state_dict(synth.classifier_spec(cfg), 500 + seed)
    cls = P.AlignmentClassifier(classifier_config=dict(params=dict(cfg)))
    cls.load_state_dict(sd)
    cls.attach(host)
    csd = ou.sub_state_dict(sd, "model.")
    g = torch.Generator().manual_seed(600 + seed)
    B, H, W, T = o["B"], o["H"], o["W"], o["T"]
    x = torch.randn(B, 4, H, W, generator=g)
    vf = torch.randn(B, T, cfg["context_dim"], generator=g)
    vf = vf / vf.norm(dim=-1, keepdim=True)
    t = torch.randint(0, 1000, (B,), generator=g).float()
    p_ref = ou.classifier_forward(csd, cfg, x, t, vf).detach()
    g_ref = osamp.classifier_grad(lambda xx, tt, cc: ou.classifier_forward(csd, cfg, xx, tt, cc), x, t, vf)
    p = cls(x.cuda(), t=t.cuda(), video_feat=vf.cuda()).cpu()
    grad, prob = host.engine.classifier_grad(x.cuda(), t.cuda(), vf.cuda(), want_prob=True)
    assert grad.shape == x.shape and torch.isfinite(grad).all(), (cfg, o)
    err = rel_l2(grad.cpu(), g_ref)
    print(f"case {seed}: {cfg} {o} -> p {p.flatten().tolist()} (ref {p_ref.flatten().tolist()}), grad rel-L2 {err:.2e}")
    assert torch.allclose(p, p_ref, atol=5e-3 * PREC_SCALE) and torch.allclose(prob.cpu(), p_ref, atol=5e-3 * PREC_SCALE), (cfg, o)
    assert err < 1.5e-2 * PREC_SCALE, (cfg, o, err)


@pytest.mark.parametrize("name,S", [("DDIM", 4), ("DPM_Solver", 4)])
def test_double_guidance_on_a_wide_latent(name, S):
    """``size_len`` = 128 (a 16 s latent, ddpm.py:1327-1356 takes any length): the classifier's first attention level then works on
    8 x 64 = 512 tokens, more than the LDS-resident backward kernels hold -- the tiled pair (csrc/backward.hip) runs inside the
    sampler.  Product (facade) against the oracle's double-guidance loop (ddim.py:344-396 / dpm_solver.py:1377-1393) with autograd
    through the oracle's classifier."""
    import diff_foley_amd as P
    from diff_foley_amd import synth
    from helpers import tiny_classifier_sd
    from oracle import unet as ou, vae as ov, samplers as osamp, schedule as osch
    sd = tiny_state_dict()
    host = P.LatentDiffusion(precision="fp16", **P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY))
    host.load_state_dict(sd)
    host.cuda()
    cls = P.AlignmentClassifier(classifier_config=dict(params=dict(synth.CLS_TINY)))
    cls.load_state_dict(tiny_classifier_sd())
    cls.attach(host)
    usd = ou.sub_state_dict(sd, "model.diffusion_model.")
    csd = ou.sub_state_dict(sd, "cond_stage_model.")
    ksd = ou.sub_state_dict(tiny_classifier_sd(), "model.")
    B, W = 1, 128
    g = torch.Generator().manual_seed(31)
    xT = torch.randn(B, 4, 16, W, generator=g)
    vf = synth.synthetic_cavp(B, 33, 64, seed=4321)
    c_ref = ov.cond_stage(csd, vf[:, :32])
    uc_ref = torch.zeros_like(c_ref)
    apply_model = lambda x, t, c: ou.unet_forward(usd, synth.UNET_TINY, x, t, c)
    classifier = lambda x, t, c: ou.classifier_forward(ksd, synth.CLS_TINY, x, t, c)
    fn = osamp.ddim_sample if name == "DDIM" else osamp.dpm_solver_sample
    z_ref, _ = fn(apply_model, osch.ddpm_schedule()["alphas_cumprod"], S, xT, c_ref, 4.5, uc_ref, classifier=classifier,
                  origin_cond=vf, classifier_scale=50.0)
    c = host.get_learned_conditioning(vf[:, :32].cuda())
    z, _ = host.sample_log_with_classifier_diff_sampler(
        c, origin_cond=vf.cuda(), batch_size=B, sampler_name=name, ddim_steps=S, size_len=W, unconditional_guidance_scale=4.5,
        unconditional_conditioning=torch.zeros_like(c), classifier=cls, classifier_guide_scale=50.0, x_T=xT.clone())
    err = rel_l2(z.cpu(), z_ref)
    print(f"double guidance on a 16 x {W} latent, {name}-{S}: rel-L2 {err:.2e}")
    assert z.shape == (B, 4, 16, W) and err < 1e-2


@pytest.mark.parametrize("W", [32, 64, 128, 192])
def test_full_classifier_gradient_on_other_latent_widths(W):
    """The FULL alignment classifier (Classifier_Backbone as configured in the reference's double-guidance YAML) on 16 x W latents:
    W = 128 / 192 put 512 / 768 tokens into its first attention level -- the tiled backward pair -- W = 32 / 64 stay on the
    LDS-resident MFMA kernel.  Gradient and probability against autograd through the oracle."""
    import diff_foley_amd as P
    from diff_foley_amd import synth
    from helpers import full_classifier_sd
    from oracle import unet as ou, samplers as osamp
    host = P.LatentDiffusion(precision="fp16", **P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY))
    host.load_state_dict(tiny_state_dict())
    host.cuda()
    sd = full_classifier_sd()
    cls = P.AlignmentClassifier(classifier_config=dict(params=dict(synth.CLS_FULL)))
    cls.load_state_dict(sd)
    cls.attach(host)
    ksd = ou.sub_state_dict(sd, "model.")
    g = torch.Generator().manual_seed(W)
    B = 2
    x = torch.randn(B, 4, 16, W, generator=g)
    vf = synth.synthetic_cavp(B, 33, 512, seed=W)
    t = torch.tensor([640.0, 12.0])
    p_ref = ou.classifier_forward(ksd, synth.CLS_FULL, x, t, vf).detach()
    g_ref = osamp.classifier_grad(lambda xx, tt, cc: ou.classifier_forward(ksd, synth.CLS_FULL, xx, tt, cc), x, t, vf)
    grad, prob = host.engine.classifier_grad(x.cuda(), t.cuda(), vf.cuda(), want_prob=True)
    err = rel_l2(grad.cpu(), g_ref)
    print(f"full classifier, 16 x {W} latent: p {prob.flatten().tolist()} (ref {p_ref.flatten().tolist()}), grad rel-L2 {err:.2e}")
    assert torch.allclose(prob.cpu(), p_ref, atol=5e-3) and err < 1e-2, (W, err)
