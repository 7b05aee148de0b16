"""Life cycle of the facade around the engine: non-default HIP streams, several models alive at once, repeated loads and moves,
creation / destruction in a loop (device memory returns), classifier guidance under a caller's stream.  Results must be the bits of
the plain single-model, default-stream run."""
import gc

import pytest
import torch

from helpers import tiny_state_dict, tiny_classifier_sd

pytestmark = pytest.mark.gpu


def _model(prec="fp16", seed=0):
    import diff_foley_amd as P
    from diff_foley_amd import synth
    m = P.LatentDiffusion(precision=prec, **P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY))
    m.load_state_dict(tiny_state_dict(seed))
    m.cuda()
    return m


def _run(m, B=2, name="DDIM", S=5):
    from diff_foley_amd import synth
    c = m.get_learned_conditioning(synth.synthetic_cavp(B, 32, 64, seed=3).cuda())
    z, _ = m.sample_log_diff_sampler(c, B, name, S, unconditional_guidance_scale=4.5, unconditional_conditioning=torch.zeros_like(c),
                                     x_T=synth.synthetic_xT(B, seed=4).cuda())
    return z, m.decode_first_stage(z)


def test_sampling_on_a_callers_stream_gives_the_default_streams_bits():
    m = _model()
    z0, d0 = _run(m)
    s = torch.cuda.Stream()
    # no wait between the two: a torch module may be called from one stream right after another (fresh activations per call); the
    # engine's plans keep workspaces across calls, so it orders the second stream behind the first itself (Engine._on) -- without
    # that the two sample() calls overlap in one workspace (NaN / garbage, seen once in three suite runs)
    for _ in range(3):
        with torch.cuda.stream(s):
            z1, d1 = _run(m)
        z2, d2 = _run(m)                                   # ... and back on the default stream, again without a wait
        s.synchronize()
        torch.cuda.current_stream().synchronize()
        assert torch.equal(z0, z1) and torch.equal(d0, d1) and torch.equal(z0, z2) and torch.equal(d0, d2)
    # classifier guidance forks a second stream off the CURRENT one and joins it again
    import diff_foley_amd as P
    from diff_foley_amd import synth
    cls = P.AlignmentClassifier(classifier_config=dict(params=dict(synth.CLS_TINY)))
    cls.load_state_dict(tiny_classifier_sd())
    cls.attach(m)
    vf = synth.synthetic_cavp(2, 33, 64, seed=9).cuda()
    c = m.get_learned_conditioning(vf[:, :32])
    kw = dict(origin_cond=vf, batch_size=2, sampler_name="DPM_Solver", ddim_steps=5, unconditional_guidance_scale=4.5,
              unconditional_conditioning=torch.zeros_like(c), classifier=cls, classifier_guide_scale=50.0,
              x_T=synth.synthetic_xT(2, seed=4).cuda())
    za, _ = m.sample_log_with_classifier_diff_sampler(c, **kw)
    s.wait_stream(torch.cuda.current_stream())             # c, vf, x_T were produced on the default stream (torch's stream rule)
    with torch.cuda.stream(s):
        zb, _ = m.sample_log_with_classifier_diff_sampler(c, **kw)
    s.synchronize()
    assert torch.equal(za, zb)


def test_two_models_alive_at_once_do_not_share_state():
    a, b = _model("fp16", 0), _model("bf16", 1)          # different weights, different libraries
    za, da = _run(a)
    zb, db = _run(b)
    za2, _ = _run(a)                                       # interleaved: a's context / plans / tables survive b's calls
    assert torch.equal(za, za2) and not torch.equal(za, zb)
    c = _model("fp16", 0)                                  # a third engine on the library a uses
    zc, dc = _run(c)
    assert torch.equal(zc, za) and torch.equal(dc, da)
    zb2, _ = _run(b)
    assert torch.equal(zb, zb2)


def test_reload_and_move_keep_results_and_engines_are_released():
    m = _model()
    z0, _ = _run(m)
    m.load_state_dict(tiny_state_dict(0))                  # same weights again: re-packed, same bits
    z1, _ = _run(m)
    m.cuda()
    m.to(torch.device("cuda", 0))
    z2, _ = _run(m)
    assert torch.equal(z0, z1) and torch.equal(z0, z2)
    m.load_state_dict(tiny_state_dict(5))                  # other weights: other result, no stale packed copy
    z3, _ = _run(m)
    assert not torch.equal(z0, z3)
    del m
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()                               # torch's cached blocks are not what is measured
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(6):                                     # engines own raw HIP allocations (weights, packed operands, plans)
        mm = _model()
        _run(mm, B=1, S=2)
        del mm
        gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 << 20, (free0, free1)        # nothing accumulates across create / destroy cycles


def test_host_threads_two_models_and_one_shared_model():
    """Two host threads, each sampling with its own model, and two threads sharing ONE model with different conditionings: every
    result is the single-threaded one, bit for bit.  (libdfengine serialises its entry points -- the calls only enqueue work -- and
    the facade lets one caller at a time run a sample() call on a model, whose context operands / timestep table / workspaces
    belong to that call.)"""
    import threading
    from diff_foley_amd import synth
    a, b = _model("fp16", 0), _model("fp16", 1)

    def job(m, seed, name, S):
        c = m.get_learned_conditioning(synth.synthetic_cavp(2, 32, 64, seed=seed).cuda())
        z, _ = m.sample_log_diff_sampler(c, 2, name, S, unconditional_guidance_scale=4.5, unconditional_conditioning=torch.zeros_like(c),
                                         x_T=synth.synthetic_xT(2, seed=seed).cuda())
        return z, m.decode_first_stage(z)
    plan = [(a, 11, "DDIM", 6), (b, 12, "DPM_Solver", 6), (a, 13, "PLMS", 5), (a, 14, "DPM_Solver", 7)]
    want = [job(*p) for p in plan]
    torch.cuda.synchronize()
    for trial in range(3):
        got = [None] * len(plan)
        errs = []

        def run(i):
            try:
                for _ in range(3):
                    got[i] = job(*plan[i])
            except Exception as e:            # surfaces in the main thread below
                errs.append((i, repr(e)))
        th = [threading.Thread(target=run, args=(i,)) for i in range(len(plan))]     # jobs 0, 2, 3 share model a; job 1 drives b
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        assert not errs, errs
        for i, ((z, d), (zw, dw)) in enumerate(zip(got, want)):
            assert torch.equal(z, zw) and torch.equal(d, dw), (trial, i)
