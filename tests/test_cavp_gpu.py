"""GPU parity of the CAVP video encoder through the C ABI (CAVPInference -> df_cavp_encode) against
(i) golden vectors produced by the reference's own ResNet3dSlowOnly / CAVP_Inference code (tests/golden/make_golden.py
--cavp; mmcv.ConvModule stand-in declared in oracle/ref_import.py) and (ii) the CPU oracle on other inputs.

Tolerance: features are L2-normalised 512-vectors; bf16 operands through 50 conv layers give rel-L2 4e-3..6e-3 (bound
1.5e-2), fp16 operands 4e-4..7e-4 (bound 2e-3).  Cosine similarity per frame > 0.9998 / 0.999995."""
import pytest
import torch

from helpers import gold, rel_l2
import diff_foley_amd as P
from diff_foley_amd import synth

pytestmark = pytest.mark.gpu

TOL = {"bf16": 1.5e-2, "fp16": 2e-3}
COS = {"bf16": 0.9998, "fp16": 0.999995}


def _model(cfg, prec):
    m = P.CAVPInference(embed_dim=cfg["embed_dim"], stage_blocks=cfg["stage_blocks"], precision=prec)
    sd = synth.make_state_dict(synth.cavp_spec(cfg))
    missing, unexpected = m.load_state_dict(sd)
    assert not missing and not unexpected
    return m.cuda(), sd


def _check(f, ref, prec, what):
    f, ref = f.cpu(), ref.cpu()
    err = rel_l2(f, ref)
    cos = torch.nn.functional.cosine_similarity(f, ref, dim=-1).min().item()
    print(f"cavp {what} [{prec}]: rel-L2 {err:.3e}  min cos {cos:.6f}")
    assert torch.isfinite(f).all()
    assert err < TOL[prec] and cos > COS[prec]


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_cavp_tiny_vs_golden_and_oracle(prec):
    from oracle import cavp as ocavp
    m, sd = _model(synth.CAVP_TINY, prec)
    g = gold("g7_cavp_tiny.npz")
    v = synth.synthetic_video(1, 4, 64, seed=77)
    _check(m.encode_video(v.cuda(), normalize=True, pool=False), g["feats"], prec, "tiny golden")
    _check(m.encode_video(v.cuda(), normalize=False, pool=False), g["feats_raw"], prec, "tiny golden (raw)")
    # two clips of an odd length and another frame size: temporal zero padding is per clip, clips are independent
    v2 = synth.synthetic_video(2, 5, 96, seed=5)
    ref = ocavp.encode_video(sd, v2, stage_blocks=tuple(synth.CAVP_TINY["stage_blocks"]))
    f2 = m.encode_video(v2.cuda(), normalize=True, pool=False)
    assert f2.shape == (2, 5, synth.CAVP_TINY["embed_dim"])
    _check(f2, ref, prec, "tiny 2x5x96 oracle")
    f1 = m.encode_video(v2[1:].cuda(), normalize=True, pool=False)
    assert torch.equal(f1[0], f2[1])            # same plan, same clip -> bit-identical
    # pool=True (cavp_model.py:58-59): MaxPool1d(16) over the frames of the projected features, then the normalisation --
    # checked against torch's own max_pool1d / normalize applied to the un-pooled features of the same clip
    with pytest.raises(RuntimeError, match="fewer than"):
        m.encode_video(v.cuda(), normalize=True, pool=True)             # 4 frames < one window of 16
    v3 = synth.synthetic_video(2, 35, 64, seed=9)
    raw = m.encode_video(v3.cuda(), normalize=False, pool=False).cpu()
    want = torch.nn.functional.max_pool1d(raw.permute(0, 2, 1), 16).squeeze(2)           # (B, C, 2): two windows, 3 frames dropped
    got = m.encode_video(v3.cuda(), normalize=False, pool=True).cpu()
    assert got.shape == want.shape == (2, synth.CAVP_TINY["embed_dim"], 2)
    assert torch.equal(got, want)
    with pytest.raises(NotImplementedError):       # the reference would normalise across the two windows there
        m.encode_video(v3.cuda(), normalize=True, pool=True)
    raw1 = m.encode_video(v3[:, :20].cuda(), normalize=False, pool=False).cpu()          # one window: squeeze(2) -> (B, C)
    for norm in (False, True):
        want = torch.nn.functional.max_pool1d(raw1.permute(0, 2, 1), 16).squeeze(2)
        if norm:
            want = torch.nn.functional.normalize(want, dim=-1)
        got = m.encode_video(v3[:, :20].cuda(), normalize=norm, pool=True).cpu()
        assert got.shape == want.shape == (2, synth.CAVP_TINY["embed_dim"])
        assert torch.allclose(got, want, rtol=0, atol=1e-6), float((got - want).abs().max())


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_cavp_full_r50_vs_golden(prec):
    m, _ = _model(synth.CAVP_FULL, prec)
    g = gold("g7_cavp_full.npz")
    v = synth.synthetic_video(1, 8, 224, seed=77)
    f = m.encode_video(v.cuda(), normalize=True, pool=False)
    assert f.shape == (1, 8, 512)
    assert torch.allclose(f.norm(dim=-1).cpu(), torch.ones(1, 8), atol=1e-4)
    _check(f, g["feats"], prec, "full R50 8x224 golden")
    _check(m.encode_video(v.cuda(), normalize=False, pool=False), g["feats_raw"], prec, "full R50 golden (raw)")


def test_cavp_autotuned_and_feeds_ldm_cond_stage():
    """Autotuned plan reproduces the golden; the features go straight into get_learned_conditioning (config 5 flow)."""
    m, _ = _model(synth.CAVP_FULL, "bf16")
    m.autotune(True)
    g = gold("g7_cavp_full.npz")
    v = synth.synthetic_video(1, 8, 224, seed=77)
    f = m.encode_video(v.cuda(), normalize=True, pool=False)
    _check(f, g["feats"], "bf16", "full R50 autotuned")
