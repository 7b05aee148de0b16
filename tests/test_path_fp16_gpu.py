"""GPU parity of the fp16-operand build (libdfengine_f16.so, ``precision="fp16"``) against the same golden vectors.

Same kernels, same speed; operands carry 11 significant bits instead of 8, so every tolerance is ~8x tighter than in
test_path_gpu.py and the north-star bound -- decoded-mel MAE < 1e-3 vs the reference CPU sampler -- holds in ABSOLUTE
mel units (the bf16 build meets it on the range-normalised mel only).  Tolerances are in the asserts."""
import numpy as np
import pytest
import torch

from helpers import (gold, rnd, rel_l2, tiny_state_dict, full_state_dict, tiny_classifier_sd, full_classifier_sd)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import diff_foley_amd
    return diff_foley_amd


@pytest.fixture(scope="module")
def tiny(P):
    from diff_foley_amd import synth
    m = P.LatentDiffusion(precision="fp16", **P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY))
    m.load_state_dict(tiny_state_dict())
    m.cuda()
    assert m.engine.precision == "fp16" and m.engine.L.df_operand_dtype() == b"f16"
    return m


@pytest.fixture(scope="module")
def full(P):
    m = P.LatentDiffusion(precision="fp16", **P.stage2_config())
    m.load_state_dict(full_state_dict())
    m.cuda()
    return m


def test_fp16_default_and_saturation_counter(P, tiny):
    """(a) the facade's default operand type is fp16, the build that meets the north-star tolerance.  (b) fp16 operands saturate
    at +-65504 (csrc/common.h op_clamp) -- silently, as far as the sample is concerned; df_debug_saturations counts, per op,
    the operand-type values stored AT the saturation point.  Procedural weights never get there (all counts 0); with one
    SpatialTransformer's GEGLU projection scaled until its outputs pass 6e4 the counter fires on that block's st.ff1 (and
    on nothing in front of it), Engine.check_saturations() raises and names the op, and the bf16 build (fp32 range) runs the
    same weights without a count."""
    from diff_foley_amd import synth, engine as E
    assert E.default_precision() == "fp16" or "DF_PRECISION" in __import__("os").environ
    cfg = P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY)
    B = 2
    x = synth.synthetic_xT(B, seed=3).cuda()
    c = tiny.get_learned_conditioning(synth.synthetic_cavp(B, 32, 64, seed=1234).cuda())
    t = torch.tensor([500.0, 37.0]).cuda()
    eng = tiny.engine
    eng.debug_saturations(True)
    try:
        y0 = tiny.apply_model(x, t, c)
        res = eng.debug_saturations_read()
        assert len(res) > 50 and all(n == 0 for _, n in res), [r for r in res if r[1]][:5]
        assert any("st.ff1" in lab for lab, _ in res) and any("groupnorm" in lab for lab, _ in res)
        eng.check_saturations()
    finally:
        eng.debug_saturations(False)
    sd = dict(tiny_state_dict())
    key = [k for k in sd if k.endswith("input_blocks.2.1.transformer_blocks.0.ff.net.0.proj.weight")]
    assert len(key) == 1
    sd[key[0]] = sd[key[0]] * 3.0e4
    runs = {}
    for prec in ("fp16", "bf16"):
        m = P.LatentDiffusion(precision=prec, **cfg)
        m.load_state_dict(sd)
        m.cuda()
        m.engine.debug_saturations(True)
        y = m.apply_model(x, t, m.get_learned_conditioning(synth.synthetic_cavp(B, 32, 64, seed=1234).cuda()))
        runs[prec] = (m, m.engine.debug_saturations_read(), y)
    m, res, y = runs["fp16"]
    hit = [(lab, n) for lab, n in res if n]
    assert hit and "st.ff1" in hit[0][0], hit[:4]          # the first op that saturates is the scaled GEGLU projection
    first = [lab for lab, _ in res].index(hit[0][0])
    assert all(n == 0 for _, n in res[:first])
    with pytest.raises(RuntimeError, match="saturated"):
        m.engine.check_saturations()
    mb, resb, yb = runs["bf16"]
    assert all(n == 0 for _, n in resb) and torch.isfinite(yb).all()
    mb.engine.check_saturations()
    assert rel_l2(y.cpu(), yb.cpu()) > 1e-2               # the clamp changed the fp16 build's result: this is what the counter is for
    assert torch.isfinite(y0).all()


def test_fp16_range_guard_on_plain_cuda(P):
    """The product's own guard (ldm.py:_range_check): load_state_dict + .cuda() runs one saturation-counted probe forward at
    t = 999 / 1.  Procedural weights: silent.  The scaled-GEGLU state dict of the test above through the PLAIN m.cuda() path:
    RuntimeWarning naming st.ff1 and the bf16 fallback; the bf16 build has no range to guard and stays silent;
    DF_RANGE_CHECK=0 switches the probe off."""
    import os
    import warnings
    from diff_foley_amd import synth
    cfg = P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        m = P.LatentDiffusion(precision="fp16", **cfg)
        m.load_state_dict(tiny_state_dict())
        m.cuda()
        assert m._range_check() == []
    sd = dict(tiny_state_dict())
    key = [k for k in sd if k.endswith("input_blocks.2.1.transformer_blocks.0.ff.net.0.proj.weight")]
    sd[key[0]] = sd[key[0]] * 3.0e4
    m = P.LatentDiffusion(precision="fp16", **cfg)
    m.load_state_dict(sd)
    with pytest.warns(RuntimeWarning, match=r"fp16 operands saturated.*st\.ff1.*precision='bf16'"):
        m.cuda()
    with pytest.warns(RuntimeWarning, match="saturated"):      # re-loading weights into a live engine probes again
        m.load_state_dict(sd)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        mb = P.LatentDiffusion(precision="bf16", **cfg)
        mb.load_state_dict(sd)
        mb.cuda()
        os.environ["DF_RANGE_CHECK"] = "0"
        try:
            m2 = P.LatentDiffusion(precision="fp16", **cfg)
            m2.load_state_dict(sd)
            m2.cuda()
        finally:
            del os.environ["DF_RANGE_CHECK"]
    # the probe leaves no state behind: the saturation counters are off and a normal forward still runs
    x = synth.synthetic_xT(2, seed=3).cuda()
    c = m.get_learned_conditioning(synth.synthetic_cavp(2, 32, 64, seed=1234).cuda())
    assert torch.isfinite(m.apply_model(x, torch.tensor([500.0, 37.0]).cuda(), c)).all()
    assert m.engine.debug_saturations_read() == []


def test_fp16_tiny_forward_and_samplers(tiny):
    from diff_foley_amd import synth
    g = gold("g3_tiny_unet.npz")
    x, c = rnd((2, 4, 16, 64), 102), rnd((2, 32, 128), 101)
    y = tiny.apply_model(x.cuda(), torch.tensor([500, 37]).cuda(), c.cuda()).cpu()
    err = rel_l2(y, g["y_int"])
    print(f"fp16 tiny UNet rel-L2 {err:.3e}")
    assert err < 3e-3
    d = tiny.decode_first_stage(rnd((2, 4, 16, 64), 103).cuda()).cpu()
    assert rel_l2(d, g["decode"]) < 3e-3
    g5 = gold("g5_tiny_samplers.npz")
    B = 2
    xT = synth.synthetic_xT(B, seed=21)
    c = tiny.get_learned_conditioning(synth.synthetic_cavp(B, 32, 64, seed=1234).cuda())
    uc = torch.zeros_like(c)
    for name, S in (("DDIM", 25), ("DPM_Solver", 25), ("PLMS", 25)):
        z, _ = tiny.sample_log_diff_sampler(c, B, name, S, unconditional_guidance_scale=4.5,
                                            unconditional_conditioning=uc, x_T=xT.clone())
        err = rel_l2(z.cpu(), g5[f"{name}_{S}_z"])
        print(f"fp16 tiny {name}-{S}: rel-L2 {err:.3e}")
        assert err < 1e-2


def test_fp16_tiny_classifier_grad(P, tiny):
    from diff_foley_amd import synth
    g = gold("g6_tiny_classifier.npz")
    cls = P.AlignmentClassifier(classifier_config=dict(params=dict(synth.CLS_TINY)))
    cls.load_state_dict(tiny_classifier_sd())
    cls.attach(tiny)
    x = rnd((2, 4, 16, 64), 105)
    vf = synth.synthetic_cavp(2, 33, 64, seed=4321)
    grad, prob = tiny.engine.classifier_grad(x.cuda(), torch.tensor([500.0, 37.0]).cuda(), vf.cuda(), want_prob=True)
    assert torch.allclose(prob.cpu(), g["cls_p"], atol=3e-3)
    err = rel_l2(grad.cpu(), g["cls_grad"])
    print(f"fp16 tiny classifier grad rel-L2 {err:.3e}")
    assert err < 1e-2


def test_fp16_full_unet_and_vae(full):
    g = gold("g4_full_unet.npz")
    x, c = rnd((2, 4, 16, 64), 200), rnd((2, 32, 768), 201)
    y = full.apply_model(x.cuda(), torch.tensor([961, 41]).cuda(), c.cuda()).cpu()
    err = rel_l2(y, g["unet_y"])
    print(f"fp16 full UNet forward: rel-L2 {err:.3e}  MAE {(y - g['unet_y']).abs().mean().item():.3e}")
    assert err < 3e-3
    d = full.decode_first_stage(rnd((1, 4, 16, 64), 202).cuda()).cpu()
    err = rel_l2(d[:, 0], g["decode"])
    print(f"fp16 full VAE decode: rel-L2 {err:.3e}")
    assert err < 3e-3


def test_fp16_full_ddim25_mel_mae_absolute(full):
    """North-star: decoded mel MAE < 1e-3 (absolute) vs the reference CPU sampler, B=1, 25-step DDIM, CFG 4.5."""
    from diff_foley_amd import synth
    g = gold("g5_full_samplers.npz")
    for seed in (21, 22):
        xT = synth.synthetic_xT(1, seed=seed)
        c = full.get_learned_conditioning(synth.synthetic_cavp(1, 32, 512, seed=1234 + seed - 21).cuda())
        uc = torch.zeros_like(c)
        z, _ = full.sample_log_diff_sampler(c, 1, "DDIM", 25, unconditional_guidance_scale=4.5,
                                            unconditional_conditioning=uc, x_T=xT.clone())
        mel = full.decode_first_stage(z)[:, 0].cpu()
        mr = g[f"ddim25_mel_{seed}"]
        mae = (mel - mr).abs().mean().item()
        print(f"fp16 seed {seed}: z rel-L2 {rel_l2(z.cpu(), g[f'ddim25_z_{seed}']):.3e}; mel MAE {mae:.3e} "
              f"(mel std {mr.std().item():.3f})")
        assert mae < 1e-3


def test_fp16_full_dpm50_and_classifier(P, full):
    from diff_foley_amd import synth
    g = gold("g5_full_samplers.npz")
    xT = synth.synthetic_xT(1, seed=21)
    c = full.get_learned_conditioning(synth.synthetic_cavp(1, 32, 512, seed=1234).cuda())
    uc = torch.zeros_like(c)
    z, _ = full.sample_log_diff_sampler(c, 1, "DPM_Solver", 50, unconditional_guidance_scale=4.5,
                                        unconditional_conditioning=uc, x_T=xT.clone())
    mel = full.decode_first_stage(z)[:, 0].cpu()
    mae = (mel - g["dpm50_mel_21"]).abs().mean().item()
    print(f"fp16 DPM-50: z rel-L2 {rel_l2(z.cpu(), g['dpm50_z_21']):.3e}; mel MAE {mae:.3e}")
    assert mae < 1e-3
    g6 = gold("g6_full_classifier.npz")
    cls = P.AlignmentClassifier(classifier_config=dict(params=dict(synth.CLS_FULL)))
    cls.load_state_dict(full_classifier_sd())
    cls.attach(full)
    x = rnd((2, 4, 16, 64), 205)
    vf = synth.synthetic_cavp(2, 33, 512, seed=4321)
    p = cls(x.cuda(), t=torch.tensor([500.0, 37.0]).cuda(), video_feat=vf.cuda()).cpu()      # keywords, as ddim.py:338 calls it
    assert torch.allclose(p, g6["cls_p"], atol=3e-3), (p, g6["cls_p"])
    grad = cls.log_prob_grad(x.cuda(), torch.tensor([500.0, 37.0]).cuda(), vf.cuda()).cpu()
    err = rel_l2(grad, g6["cls_grad"])
    print(f"fp16 full classifier grad rel-L2 {err:.3e}")
    assert err < 1e-2


def test_unrequested_precision_moves_itself_to_bf16_when_fp16_saturates(P, monkeypatch):
    """No operand type requested (the notebook's case): the product picks fp16 AND keeps it safe.  The scaled-GEGLU state dict that
    saturates fp16 (tests above) through a plain LatentDiffusion(**cfg) -> load_state_dict -> .cuda(): one RuntimeWarning, the model
    is on the bf16 build afterwards, samples are finite and equal the explicit bf16 model's bit for bit; procedural weights stay on
    fp16 silently; an explicit precision='fp16' is obeyed (warning only, covered above)."""
    import warnings
    from diff_foley_amd import synth
    monkeypatch.delenv("DF_PRECISION", raising=False)
    cfg = P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        m0 = P.LatentDiffusion(**cfg)
        m0.load_state_dict(tiny_state_dict())
        m0.cuda()
    assert m0.engine.precision == "fp16"
    sd = dict(tiny_state_dict())
    key = [k for k in sd if k.endswith("input_blocks.2.1.transformer_blocks.0.ff.net.0.proj.weight")]
    sd[key[0]] = sd[key[0]] * 3.0e4
    m = P.LatentDiffusion(**cfg)
    m.load_state_dict(sd)
    with pytest.warns(RuntimeWarning, match=r"now runs on the bf16 build"):
        m.cuda()
    assert m.engine.precision == "bf16" and m.precision == "bf16"
    mb = P.LatentDiffusion(precision="bf16", **cfg)
    mb.load_state_dict(sd)
    mb.cuda()
    B = 2
    xT = synth.synthetic_xT(B, seed=21).cuda()
    outs = []
    for mm in (m, mb):
        c = mm.get_learned_conditioning(synth.synthetic_cavp(B, 32, 64, seed=1234).cuda())
        z, _ = mm.sample_log_diff_sampler(c, B, "DDIM", 4, unconditional_guidance_scale=4.5, unconditional_conditioning=torch.zeros_like(c),
                                          x_T=xT.clone())
        outs.append(z)
    assert torch.isfinite(outs[0]).all()
    assert rel_l2(outs[0].cpu(), outs[1].cpu()) < 1e-2
    assert torch.equal(outs[0], outs[1])


def test_requant_debug_hook_emulates_the_bf16_build(P, tiny):
    """df_debug_requant (tools/error_budget.py): in the fp16 build, re-rounding the operand-type outputs of every op to bf16 precision
    moves the result away from the plain fp16 result by about the distance of the bf16 build, re-rounding nothing changes nothing,
    and the hook leaves no state behind."""
    from diff_foley_amd import synth
    B = 2
    x = synth.synthetic_xT(B, seed=3).cuda()
    c = tiny.get_learned_conditioning(synth.synthetic_cavp(B, 32, 64, seed=1234).cuda())
    t = torch.tensor([500.0, 37.0]).cuda()
    eng = tiny.engine
    y0 = tiny.apply_model(x, t, c).clone()
    eng.debug_requant("")
    assert torch.equal(tiny.apply_model(x, t, c), y0)
    eng.debug_requant("*")
    y_all = tiny.apply_model(x, t, c).clone()
    eng.debug_requant("groupnorm")
    y_gn = tiny.apply_model(x, t, c).clone()
    eng.debug_requant("")
    assert torch.equal(tiny.apply_model(x, t, c), y0)
    d_all, d_gn = rel_l2(y_all.cpu(), y0.cpu()), rel_l2(y_gn.cpu(), y0.cpu())
    assert 1e-4 < d_gn <= d_all * 1.5 and 5e-4 < d_all < 5e-2, (d_gn, d_all)
