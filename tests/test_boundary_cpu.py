"""CPU: the drop-in boundary's host-side contract (no GPU): reference kwargs that the path does not implement raise
instead of being swallowed, the classifier mirrors the reference argument order, unsupported UNet variants are refused
at construction, and the samplers' guards fire before any device work."""
import inspect

import pytest
import torch

import diff_foley_amd as P
from diff_foley_amd import samplers as S, synth


@pytest.mark.parametrize("kw", [dict(quantize_x0=True)])
def test_unsupported_sampler_kwargs_raise(kw):
    """ddim.py:58-113 accepts this; it selects the VQ first stage's quantiser, which an AutoencoderKL model does not have.  (mask /
    x0 -- inpainting -- and the score_corrector callback ARE on the path since round 3, noise_dropout since round 5:
    tests/test_path_gpu.py::test_tiny_inpainting_vs_golden, ::test_score_corrector_callback,
    ::test_tiny_stochastic_ddim_vs_golden.)"""
    with pytest.raises(NotImplementedError):
        S.reject_unsupported("DDIMSampler", kw)


def test_dpm_solver_refuses_the_mask_its_reference_drops():
    """dpm_solver/sampler.py:24-56 takes mask / x0 and never uses them: an inpainting request would silently come back as a plain
    sample, so it raises here."""
    with pytest.raises(NotImplementedError):
        S.reject_unsupported("DPMSolverSampler", dict(mask=torch.ones(1, 1, 16, 64)), dict(mask=None, x0=None))


def test_default_and_unknown_kwargs_pass():
    S.reject_unsupported("DDIMSampler", dict(mask=None, x0=None, quantize_x0=False, noise_dropout=0.0, score_corrector=None,
                                             some_future_flag=3, verbose=False))


def test_classifier_mirrors_reference_argument_order():
    """alignment_classifier.py:269  forward(self, spec_noisy, video_feat, t)."""
    sig = list(inspect.signature(P.AlignmentClassifier.forward).parameters)
    assert sig == ["self", "spec_noisy", "video_feat", "t"]
    cls = P.AlignmentClassifier(classifier_config=dict(params=dict(synth.CLS_TINY)))
    with pytest.raises(RuntimeError, match="not attached"):
        cls(torch.zeros(1, 4, 16, 64), video_feat=torch.zeros(1, 33, 64), t=torch.zeros(1))


@pytest.mark.parametrize("bad", [dict(transformer_depth=2), dict(num_head_channels=64), dict(legacy=True),
                                 dict(use_scale_shift_norm=True), dict(resblock_updown=True), dict(dims=3),
                                 dict(use_spatial_transformer=False), dict(num_classes=10)])
def test_unsupported_unet_variants_are_refused(bad):
    cfg = P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY)
    cfg["unet_config"]["params"].update(bad)
    with pytest.raises(NotImplementedError):
        P.LatentDiffusion(**cfg)
    cc = dict(params=dict(synth.CLS_TINY, **bad))
    with pytest.raises(NotImplementedError):
        P.AlignmentClassifier(classifier_config=cc)


@pytest.mark.parametrize("bad", [dict(attn_resolutions=[16]), dict(resamp_with_conv=False), dict(tanh_out=True),
                                 dict(give_pre_end=True), dict(use_linear_attn=True), dict(attn_type="linear"), dict(dropout=0.1)])
def test_unsupported_decoder_variants_are_refused(bad):
    """Decoder constructor arguments (stage1_autoencoder/model.py:557-561) that select code which is not built: refused when the
    model is constructed, never ignored (an ignored ``attn_resolutions`` would decode with a different network)."""
    cfg = P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY)
    cfg["first_stage_config"]["params"]["ddconfig"].update(bad)
    with pytest.raises(NotImplementedError):
        P.LatentDiffusion(**cfg)


def test_reference_yaml_defaults_are_accepted():
    cfg = P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY)
    cfg["unet_config"]["params"].update(dict(transformer_depth=1, legacy=False, use_checkpoint=True, image_size=32,
                                             dropout=0.0, use_spatial_transformer=True))
    m = P.LatentDiffusion(**cfg)
    assert m.num_timesteps == 1000 and m.channels == 4


def test_engine_rejects_wrong_feature_width_before_touching_the_device():
    from diff_foley_amd import engine as E
    with pytest.raises(RuntimeError, match="expects 768"):
        E._want_dim("cross-attention context", 512, 768)
    E._want_dim("x", 768, 768)
    E._want_dim("x", 5, None)


def test_a_checkpoint_of_another_architecture_is_a_size_mismatch_error():
    """torch's load_state_dict raises on a size mismatch whatever ``strict`` says; the engine's packers take each tensor's own shape,
    so without this check a checkpoint of another architecture loads and samples as a different network (found by
    tools/load_probe.py: a conv weight with 63 input channels, a bias of twice the length)."""
    spec = synth.state_dict_spec(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY)
    sd = {k: torch.zeros(*shape) for k, shape in spec.items()}
    m = P.LatentDiffusion(**P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY))
    assert m.load_state_dict(dict(sd, **{"first_stage_model.encoder.conv_in.weight": torch.zeros(3), "loss.x": torch.zeros(1)}))[0] == []
    k = "model.diffusion_model.input_blocks.1.0.in_layers.2.weight"
    for bad in (sd[k][:, :-1], sd[k].reshape(-1), torch.zeros(2, *sd[k].shape)):
        with pytest.raises(RuntimeError, match="size mismatch for " + k.replace(".", r"\.")):
            m.load_state_dict(dict(sd, **{k: bad}))
    with pytest.raises(RuntimeError, match="size mismatch"):
        m.load_state_dict(dict(sd, **{"cond_stage_model.pos_emb.weight": torch.zeros(41, 128)}))
    with pytest.raises(RuntimeError, match="size mismatch"):
        m.load_state_dict(dict(sd, **{"first_stage_model.decoder.conv_out.bias": torch.zeros(1)}))
    csd = {k: torch.zeros(*shape) for k, shape in synth.classifier_spec(synth.CLS_TINY).items()}
    cls = P.AlignmentClassifier(classifier_config=dict(params=dict(synth.CLS_TINY)))
    cls.load_state_dict(csd)
    with pytest.raises(RuntimeError, match="size mismatch"):
        cls.load_state_dict(dict(csd, **{"model.classifier.weight": torch.zeros(2, 64)}))
    vsd = {k: torch.zeros(*shape) for k, shape in synth.cavp_spec(synth.CAVP_TINY).items()}
    enc = P.CAVPInference(embed_dim=synth.CAVP_TINY["embed_dim"], stage_blocks=synth.CAVP_TINY["stage_blocks"])
    assert enc.load_state_dict(vsd) == ([], [])
    with pytest.raises(RuntimeError, match="size mismatch"):
        enc.load_state_dict(dict(vsd, **{"video_project_head.weight": torch.zeros(64, 100)}))
    with pytest.warns(RuntimeWarning, match="non-finite"):
        m.load_state_dict(dict(sd, **{k: torch.full_like(sd[k], float("nan"))}))
