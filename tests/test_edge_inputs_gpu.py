"""Edge inputs at the drop-in boundary.  The C ABI takes raw pointers and sizes, so everything a torch module would have rejected
(or broadcast) by shape has to be rejected (or broadcast) before the call: an empty batch must be an exception, not a division by
zero inside a plan builder (round 6: SIGFPE of the whole interpreter before); a timestep vector shorter than the batch must not be read
out of bounds; a tensor with the wrong channel count must not be re-interpreted.  Inputs torch would have accepted as they are --
non-contiguous, float64, on the host -- give the result of the contiguous fp32 device tensor."""
import pytest
import torch

from helpers import rel_l2, tiny_state_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny():
    import diff_foley_amd as P
    from diff_foley_amd import synth
    m = P.LatentDiffusion(precision="fp16", **P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY))
    m.load_state_dict(tiny_state_dict())
    m.cuda()
    return m


@pytest.fixture(scope="module")
def cond(tiny):
    from diff_foley_amd import synth
    return tiny.get_learned_conditioning(synth.synthetic_cavp(2, 32, 64).cuda())


def test_empty_batches_give_empty_results_and_empty_maps_raise(tiny, cond):
    """A torch module takes a zero-row batch and returns a zero-row result (conv / GroupNorm / attention all do); so does the facade
    -- without a plan, without a launch -- which is what a rank with an empty shard of a small global batch needs
    (parallel.shard_range).  Zero-sized MAPS have no reference behaviour worth keeping: they raise."""
    z = lambda *s: torch.zeros(*s).cuda()
    assert tiny.get_learned_conditioning(z(0, 32, 64)).shape == (0, 32, 128)
    assert tiny.get_learned_conditioning(z(2, 0, 64)).shape == (2, 0, 128)
    assert tiny.decode_first_stage(z(0, 4, 16, 64)).shape == (0, 3, 64, 256)
    assert tiny.apply_model(z(0, 4, 16, 64), z(0), cond[:0]).shape == (0, 4, 16, 64)
    for name in ("DDIM", "PLMS", "DPM_Solver"):
        zz, inter = tiny.sample_log_diff_sampler(cond[:0], 0, name, 4, unconditional_guidance_scale=4.5,
                                                 unconditional_conditioning=cond[:0])
        assert zz.shape == (0, 4, 16, 64)
        assert inter is None or all(t.shape == (0, 4, 16, 64) for t in inter["x_inter"])
    zz, inter = tiny.sample(cond[:0], batch_size=0, return_intermediates=True, timesteps=3, shape=(0, 4, 16, 64))
    assert zz.shape == (0, 4, 16, 64) and len(inter) >= 1
    with pytest.raises(RuntimeError, match="must be positive"):
        tiny.decode_first_stage(z(1, 4, 0, 64))
    with pytest.raises(RuntimeError, match="must be positive"):
        tiny.apply_model(z(1, 4, 0, 64), z(1), cond[:1])
    with pytest.raises(RuntimeError, match="must be positive"):          # a context without tokens: softmax over nothing
        tiny.apply_model(z(2, 4, 16, 64), z(2), torch.zeros(2, 0, 128).cuda())
    # straight at the C ABI (a binding that does not check): an exception with a message, never a signal
    eng = tiny.engine
    from diff_foley_amd.engine import _chk, _ptr, _stream
    out = z(1)
    with pytest.raises(RuntimeError, match="must be positive"):
        _chk(eng.L.df_cond_encode(eng._h, _ptr(out), _ptr(out), 0, 32, _stream()), eng.L)
    with pytest.raises(RuntimeError, match="must be positive"):
        _chk(eng.L.df_vae_decode(eng._h, _ptr(out), _ptr(out), 0, 16, 64, _stream()), eng.L)
    with pytest.raises(RuntimeError, match="must be positive|UNet batch is 0"):
        _chk(eng.L.df_unet_forward(eng._h, _ptr(out), _ptr(out), _ptr(out), 0, 16, 64, _stream()), eng.L)
    # the model is still usable afterwards
    assert torch.isfinite(tiny.decode_first_stage(z(1, 4, 16, 64))).all()


def test_step_counts_at_the_edges_behave_like_the_reference(tiny, cond):
    with pytest.raises(ZeroDivisionError):                     # util.py:48: c = num_ddpm_timesteps // num_ddim_timesteps
        tiny.sample_log_diff_sampler(cond, 2, "DDIM", 0)
    with pytest.raises(IndexError):                            # S = 1000: timestep 999 + 1 (util.py:57)
        tiny.sample_log_diff_sampler(cond, 2, "DDIM", 1000)
    with pytest.raises(AssertionError):                        # dpm_solver.py:1083: steps >= order
        tiny.sample_log_diff_sampler(cond, 2, "DPM_Solver", 1)
    from diff_foley_amd import synth
    from oracle import unet as ou, samplers as osamp, schedule as osch
    usd = ou.sub_state_dict(tiny_state_dict(), "model.diffusion_model.")
    apply_model = lambda x, t, c: ou.unet_forward(usd, synth.UNET_TINY, x, t, c)
    xT = synth.synthetic_xT(2, seed=5)
    acp = osch.ddpm_schedule()["alphas_cumprod"]
    for name, fn in (("DDIM", osamp.ddim_sample), ("PLMS", osamp.plms_sample)):         # one step: timesteps = [1]
        z, inter = tiny.sample_log_diff_sampler(cond, 2, name, 1, x_T=xT.clone())
        z_ref, inter_ref = fn(apply_model, acp, 1, xT, cond.cpu())
        assert len(inter["x_inter"]) == len(inter_ref["x_inter"]) == 2
        assert rel_l2(z.cpu(), z_ref) < 5e-3, name


def test_timestep_vector_is_broadcast_or_rejected_never_overread(tiny, cond):
    x = torch.randn(2, 4, 16, 64, generator=torch.Generator().manual_seed(1)).cuda()
    full = tiny.apply_model(x, torch.tensor([37.0, 37.0]).cuda(), cond)
    one = tiny.apply_model(x, torch.tensor([37.0]).cuda(), cond)               # the reference's embedding broadcasts over the batch
    zero_d = tiny.apply_model(x, torch.tensor(37).cuda(), cond)
    assert torch.equal(full, one) and torch.equal(full, zero_d)
    with pytest.raises(RuntimeError, match="timesteps"):
        tiny.apply_model(x, torch.tensor([1.0, 2.0, 3.0]).cuda(), cond)
    tiny.engine.set_context(torch.cat([torch.zeros_like(cond), cond]))
    a = tiny.engine.unet_forward_cfg(x, torch.tensor([500.0, 500.0]).cuda(), 4.5)
    b = tiny.engine.unet_forward_cfg(x, torch.tensor([500.0]).cuda(), 4.5)
    assert torch.equal(a, b)
    tiny._ctx_owner = None


def test_wrong_ranks_and_channel_counts_are_rejected(tiny, cond):
    t = torch.tensor([5.0, 6.0]).cuda()
    with pytest.raises(RuntimeError, match="channels"):
        tiny.apply_model(torch.randn(2, 3, 16, 64).cuda(), t, cond)
    with pytest.raises(RuntimeError, match="4-D"):
        tiny.apply_model(torch.randn(2, 4, 16 * 64).cuda(), t, cond)
    with pytest.raises(RuntimeError, match="channels"):
        tiny.decode_first_stage(torch.randn(2, 8, 16, 64).cuda())
    with pytest.raises(RuntimeError, match="not divisible"):
        tiny.apply_model(torch.randn(2, 4, 15, 64).cuda(), t, cond)
    with pytest.raises(RuntimeError, match="context has 1 rows"):
        tiny.apply_model(torch.randn(2, 4, 16, 64).cuda(), t, cond[:1])
    with pytest.raises(RuntimeError, match="last dimension"):
        tiny.apply_model(torch.randn(2, 4, 16, 64).cuda(), t, torch.randn(2, 32, 96).cuda())
    with pytest.raises(RuntimeError, match="last dimension"):
        tiny.get_learned_conditioning(torch.randn(2, 32, 63).cuda())


def test_layouts_dtypes_and_devices_torch_would_accept(tiny, cond):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 4, 16, 64, generator=g)
    t = torch.tensor([961, 1])
    want = tiny.apply_model(x.cuda(), t.cuda(), cond)
    nhwc = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)             # same values, channels-last strides
    assert not nhwc.is_contiguous()
    assert torch.equal(tiny.apply_model(nhwc.cuda(), t.cuda(), cond), want)
    assert torch.equal(tiny.apply_model(x.double().cuda(), t.cuda(), cond), want)
    assert torch.equal(tiny.apply_model(x, t, cond.cpu()), want)                # host tensors are moved, like module(x.to(device))
    assert torch.equal(tiny.apply_model(x.cuda(), t.to(torch.int32).cuda(), cond), want)
    zd = tiny.decode_first_stage(x.cuda())
    assert torch.equal(tiny.decode_first_stage(nhwc.double()), zd)
    # timesteps far outside the trained range stay finite (sinusoidal embedding)
    assert torch.isfinite(tiny.apply_model(x.cuda(), torch.tensor([1e6, -5.0]).cuda(), cond)).all()


def test_nan_input_propagates_on_the_bf16_build():
    """The reference (fp32 torch) turns a NaN latent into a NaN output.  So does the bf16-operand build.  The fp16-operand build's
    operand stores SATURATE (hardware clamp at +-65504) and a saturating conversion maps NaN to a finite value:
    there a NaN input comes out finite -- stated here, not asserted as a feature; its range probe counts the saturated stores."""
    import diff_foley_amd as P
    from diff_foley_amd import synth
    m = P.LatentDiffusion(precision="bf16", **P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY))
    m.load_state_dict(tiny_state_dict())
    m.cuda()
    c = m.get_learned_conditioning(synth.synthetic_cavp(2, 32, 64).cuda())
    x = torch.randn(2, 4, 16, 64)
    x[1, 2, 3, 4] = float("nan")
    y = m.apply_model(x.cuda(), torch.tensor([961, 1]).cuda(), c)
    assert torch.isnan(y[1]).any() and torch.isfinite(y[0]).all()              # samples do not mix
