"""Sampler host loops of the MI355X path: DDIM, DPM-Solver++(2M), PLMS, ancestral DDPM.

Same call surface as the reference sampler classes (``DDIMSampler(model).sample(S, batch_size, shape,
conditioning, ...) -> (samples, intermediates)``), but every tensor operation in the loop is a HIP
kernel behind the C ABI: the CFG UNet evaluation (``df_unet_forward_cfg``: batch duplication, UNet,
guidance combine) and the solver updates (``df_ddim_update`` / ``df_lincomb``).  The cross-attention
context is handed to the engine once per ``sample()`` (it is step-invariant), so the K/V projections
of the 16 SpatialTransformers leave the step loop.  torch is used for RNG (x_T, eta noise) only.

Reference: diff_foley/models/diffusion/ddim.py:58-273, plms.py:60-236,
dpm_solver/sampler.py:24-156 + dpm_solver/dpm_solver.py:504-549,755-810,1071-1105, ddpm.py:1083-1268.
"""
import numpy as np
import warnings

import torch

from . import engine as E
from .schedule import DDIMTables, DPMTables

import os as _os
_NO_HOIST = bool(int(_os.environ.get("DF_NO_THOIST", "0")))      # A/B switch: time embedding inside every step


def _warn_batch(conditioning, batch_size):
    if conditioning is not None:
        c = conditioning[list(conditioning.keys())[0]] if isinstance(conditioning, dict) else conditioning
        if isinstance(c, (list, tuple)):
            c = c[0]
        if c.shape[0] != batch_size:
            print(f"Warning: Got {c.shape[0]} conditionings but batch-size is {batch_size}")


# Reference sampler features that are NOT on this path.  The reference classes accept them (ddim.py:58-113, 179-273;
# plms.py:60-110; dpm_solver/sampler.py:24-56); silently dropping one would return a plausible but wrong sample, so a
# non-default value raises.  Anything not listed is ignored exactly like the reference's own **kwargs.
# (`normals_sequence` is accepted and never read by the reference's samplers -- ddim.py:65,123, plms.py:64 -- so it is ignored here too.)
# (`noise_dropout` IS on the path since round 5: DDIMSampler applies it as ddim.py:270-271 does; with PLMS -- eta = 0, no noise --
# and DPM-Solver++ -- whose reference sampler drops the argument -- it has nothing to act on and is accepted like the reference.)
_UNSUPPORTED_DEFAULTS = dict(quantize_x0=False)


def reject_unsupported(sampler, kwargs, extra=None):
    checks = dict(_UNSUPPORTED_DEFAULTS)
    checks.update(extra or {})
    for k, default in checks.items():
        if k in kwargs and kwargs[k] is not None and kwargs[k] != default:
            raise NotImplementedError(f"{sampler}: {k}={kwargs[k]!r} is not supported by the MI355X sampling path "
                                      f"(only the default {default!r}); see DESIGN.md section 1")


def _is_table_error(ex):
    """The engine's two messages for a missing / outgrown hoisted timestep table (csrc/engine.hip unet_forward_ts)."""
    msg = str(ex)
    return "no timestep table" in msg or "outside the table" in msg or "hoistable" in msg


class _Inpaint:
    """mask / x0 of the reference samplers: the known region is re-noised to the current step and pasted over the sample,
    img <- q_sample(x0, t) * mask + (1 - mask) * img  (ddim.py:206-209, plms.py:147-150, ddpm.py:1239-1241; q_sample ddpm.py:279-282).
    q_noise_fn(shape) -> tensor replaces torch.randn_like(x0) (tests: the reference's noise sequence)."""

    def __init__(self, model, mask, x0, size, q_noise_fn=None):
        self.on = mask is not None
        if not self.on:
            return
        assert x0 is not None                                  # like the reference
        dev = model.device
        B, C, H, W = size
        x0 = x0.to(dev, torch.float32)
        mask = mask.to(dev, torch.float32)
        if mask.dim() == 3:
            mask = mask[:, None]
        if mask.dim() != 4 or x0.dim() != 4 or mask.shape[1] not in (1, C) or tuple(mask.shape[2:]) not in ((H, W), (1, 1)) or \
                mask.shape[0] not in (1, B) or tuple(x0.shape[1:]) != (C, H, W) or x0.shape[0] not in (1, B):
            raise ValueError(f"inpainting: mask {tuple(mask.shape)} / x0 {tuple(x0.shape)} do not broadcast to the latent "
                             f"{(B, C, H, W)} (mask [B][1|C][H][W], x0 [B][C][H][W])")
        mc = C if mask.shape[1] == C and C != 1 else 1
        self.mask = mask.expand(B, mc, H, W).contiguous()      # broadcast copy, no arithmetic
        self.x0 = x0.expand(B, C, H, W).contiguous()
        self.sa = model.sqrt_alphas_cumprod.cpu().numpy()
        self.s1m = model.sqrt_one_minus_alphas_cumprod.cpu().numpy()
        self.noise_fn = q_noise_fn

    def blend(self, img, t):
        noise = self.noise_fn(tuple(self.x0.shape)).to(img.device, torch.float32) if self.noise_fn is not None \
            else torch.randn_like(self.x0)
        return E.q_sample_blend(img, self.x0, noise.contiguous(), self.mask, self.sa[int(t)], self.s1m[int(t)])


class _Guided:
    """eps(x, t) with classifier-free guidance; owns the engine context for the duration of a sample().

    ``timesteps`` (optional): every timestep value the sampler is going to visit, in its own indexing.  The time embedding
    (timestep_embedding -> time_embed MLP -> the emb projection of all 22 ResBlocks; util.py:151-171,
    openai_unetmodel.py:506-511,262) depends on t only, so it is computed for all of them here, before the loop
    (``df_unet_set_timesteps``), and ``eps_fn(x, t, k)`` takes table row ``k`` instead of four launches per step."""

    def __init__(self, model, cond, scale, uc, timesteps=None, size=None):
        self.m = model
        self.eng = model.engine
        self.cfg = not (uc is None or scale == 1.0)
        self.scale = float(scale)
        cond = model._cond_tensor(cond)
        self._ctx = torch.cat([model._cond_tensor(uc), cond]) if self.cfg else cond
        self.eng.set_context(self._ctx)
        model._ctx_owner = None          # invalidate apply_model's cached context
        self.hoisted = False
        if timesteps is not None and size is not None and not _NO_HOIST:
            B, _, H, W = size
            try:      # an optimisation, never a requirement: a UNet configuration without a hoistable time embedding
                self.eng.set_timesteps([float(v) for v in timesteps], B, H, W, self.cfg)      # keeps the in-step t path
                self.hoisted = True
            except RuntimeError as ex:
                if not _is_table_error(ex):      # anything else (bad shapes, a sticky HIP error) is the caller's to see
                    raise
                self.hoisted = False

    def reclaim_context(self):
        """A caller-supplied callback (score_corrector) may have evaluated the model itself -- ``model.apply_model`` and the
        ``model.model`` facades set THEIR context in the engine.  Put the sampler's [uncond ; cond] context back."""
        if self.m._ctx_owner is not None:
            self.eng.set_context(self._ctx)
            self.m._ctx_owner = None

    def __call__(self, x, t, k=None):
        k = k if self.hoisted else None
        fwd = (lambda kk: self.eng.unet_forward_cfg(x, t, self.scale, ts_index=kk)) if self.cfg else \
              (lambda kk: self.eng.unet_forward(x, t, ts_index=kk))
        if k is None:
            return fwd(None)
        try:
            return fwd(k)
        except RuntimeError as ex:
            # the plan that owned the timestep table was rebuilt in the middle of the sample (plan-cache eviction, a callback that
            # re-finalised the engine): the in-step path computes the same embedding from t, bit-identically.  ONLY that case
            # falls back (the engine names it in its message); every other engine error propagates.
            if not _is_table_error(ex):
                raise
            warnings.warn("diff_foley_amd: the hoisted time-embedding table was dropped in the middle of a sample() call "
                          f"({ex}); the remaining steps compute the embedding inside the step", RuntimeWarning, stacklevel=2)
            self.hoisted = False
            return fwd(None)


def _classifier_grad(model, classifier, x, t, origin_cond):
    if not hasattr(classifier, "log_prob_grad"):
        raise RuntimeError("classifier guidance needs a diff_foley_amd AlignmentClassifier (log_prob_grad)")
    if classifier.engine is None:        # the notebook passes a freshly loaded classifier (ipynb:288-311)
        classifier.attach(model)
    return classifier.log_prob_grad(x, t, origin_cond)


_side_streams = {}


def _eps_and_classifier_grad(model, classifier, eps_call, x, t, origin_cond):
    """eps(x, t) and the classifier's d log p / d x at the same (x, t) (ddim.py:374-380, dpm_solver.py:1377-1393) do not depend on
    each other, so the gradient's launches go to a second HIP stream beside the UNet step's (round 6).  The gradient plan is ~160
    launches of small grids at the launch floor (1.14 ms per call at B = 8, profiles/r6_classifier_grad_ops.txt); beside the UNet step
    they run on CUs the step's kernels leave idle: configs[2] 152.5 -> 171.1 steps/s, latents bit-equal to the one-stream order
    (tools/cls_overlap_probe.py, profiles/r6_classifier_overlap.txt).  The plans own their workspaces; the only shared inputs are x and
    t, read-only in both.  (Two UNet chains do NOT overlap this way -- every one of their kernels wants the whole chip:
    tools/two_chain_probe.py.)  DF_CLS_OVERLAP=0 restores the one-stream order.  Returns (eps, grad), both ready on the current stream."""
    if _os.environ.get("DF_CLS_OVERLAP", "1") == "0":
        return eps_call(), _classifier_grad(model, classifier, x, t, origin_cond)
    cur = torch.cuda.current_stream(x.device)
    side = _side_streams.get(x.device)
    if side is None:
        side = _side_streams[x.device] = torch.cuda.Stream(x.device)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        g = _classifier_grad(model, classifier, x, t, origin_cond)
    x.record_stream(side)
    t.record_stream(side)
    e = eps_call()
    cur.wait_stream(side)
    g.record_stream(cur)
    return e, g


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0.0, verbose=True):
        # "uniform" | "quad" (util.py:46-60).  As in the reference, sample() re-makes the schedule with the default ("uniform",
        # ddim.py:90): a "quad" schedule serves callers that drive the step loop from the tables themselves.
        self.tables = DDIMTables(self.model.alphas_cumprod, ddim_num_steps, ddim_eta, ddim_discretize)
        self.ddim_timesteps = self.tables.timesteps
        self.ddim_alphas = self.tables.alphas
        self.ddim_alphas_prev = self.tables.alphas_prev
        self.ddim_sigmas = self.tables.sigmas
        self.ddim_sqrt_one_minus_alphas = self.tables.sqrt_one_minus_alphas

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, eta=0.0, x_T=None, log_every_t=100,
               unconditional_guidance_scale=1.0, unconditional_conditioning=None, temperature=1.0,
               verbose=True, callback=None, img_callback=None, classifier=None, origin_cond=None,
               classifier_guide_scale=0.0, mask=None, x0=None, score_corrector=None, corrector_kwargs=None,
               noise_dropout=0.0, **kwargs):
        reject_unsupported("DDIMSampler", kwargs)
        noise_fn = kwargs.get("noise_fn")       # tests: replaces the device generator (the reference run's CPU noise sequence)
        _warn_batch(conditioning, batch_size)
        self.make_schedule(S, ddim_eta=eta, verbose=verbose)
        dev = self.model.device
        C, H, W = shape
        size = (batch_size, C, H, W)
        img = torch.randn(size, device=dev) if x_T is None else x_T.to(dev, torch.float32).contiguous()
        tb = self.tables
        steps = np.flip(tb.timesteps)
        total = steps.shape[0]
        eps_fn = _Guided(self.model, conditioning, unconditional_guidance_scale, unconditional_conditioning, steps, size)
        t_all = torch.tensor(steps.copy(), dtype=torch.float32, device=dev)[:, None].expand(total, batch_size).contiguous()
        t_long = t_all.long() if score_corrector is not None else None     # the reference hands the callback `ts` (int64, ddim.py:217)
        inter = {"x_inter": [img], "pred_x0": [img]}
        paint = _Inpaint(self.model, mask, x0, size, kwargs.get("q_noise_fn"))
        for i in range(total):
            index = total - i - 1
            if paint.on:                     # ddim.py:206-209
                img = paint.blend(img, steps[i])
            a_t, a_prev = tb.alphas[index], tb.alphas_prev[index]
            if classifier is None:
                e_t = eps_fn(img, t_all[i], i)
            else:                            # ddim.py:374-380: the classifier gradient first ...
                e_t, g = _eps_and_classifier_grad(self.model, classifier, lambda: eps_fn(img, t_all[i], i), img, t_all[i], origin_cond)
                e_t = E.lincomb([(1.0, e_t), (-np.sqrt(np.float32(1.0) - a_t) * classifier_guide_scale, g)])
            if score_corrector is not None:  # ... then the caller's callback on the guided eps (ddim.py:249-251, 382-384)
                e_t = score_corrector.modify_score(self.model, e_t, img, t_long[i], conditioning, **(corrector_kwargs or {}))
                e_t = E._dev_f32(e_t, dev)
                eps_fn.reclaim_context()
            sigma = tb.sigmas[index]
            noise = None
            if sigma != 0.0:         # ddim.py:269-271: noise * temperature, then dropout of the noise (functional default: training)
                noise = (noise_fn(size) if noise_fn is not None else torch.randn(size, device=dev)) * temperature
                if noise_dropout > 0.0:
                    noise = torch.nn.functional.dropout(noise, p=float(noise_dropout))
                noise = noise.to(dev, torch.float32).contiguous()
            img, pred_x0 = E.ddim_update(img, e_t, a_t, a_prev, sigma, tb.sqrt_one_minus_alphas[index], noise)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total - 1:
                inter["x_inter"].append(img)
                inter["pred_x0"].append(pred_x0)
        return img, inter

    def sample_with_classifier(self, S, batch_size, shape, conditioning=None, origin_cond=None, classifier=None,
                               classifier_guide_scale=0.0, **kwargs):
        return self.sample(S, batch_size, shape, conditioning, origin_cond=origin_cond, classifier=classifier,
                           classifier_guide_scale=classifier_guide_scale, **kwargs)


class PLMSSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, eta=0.0, x_T=None, log_every_t=100,
               unconditional_guidance_scale=1.0, unconditional_conditioning=None, verbose=True, callback=None,
               img_callback=None, mask=None, x0=None, score_corrector=None, corrector_kwargs=None, **kwargs):
        if eta != 0:
            raise ValueError("ddim_eta must be 0 for PLMS")
        reject_unsupported("PLMSSampler", kwargs, dict(temperature=1.0))
        _warn_batch(conditioning, batch_size)
        tb = DDIMTables(self.model.alphas_cumprod, S, 0.0)
        dev = self.model.device
        C, H, W = shape
        size = (batch_size, C, H, W)
        img = torch.randn(size, device=dev) if x_T is None else x_T.to(dev, torch.float32).contiguous()
        steps = np.flip(tb.timesteps)
        total = steps.shape[0]
        guided = _Guided(self.model, conditioning, unconditional_guidance_scale, unconditional_conditioning, steps, size)
        t_all = torch.tensor(steps.copy(), dtype=torch.float32, device=dev)[:, None].expand(total, batch_size).contiguous()
        t_long = t_all.long() if score_corrector is not None else None     # the reference's `ts` / `ts_next` are int64 (plms.py:140-142)

        def eps_fn(x, k):                    # get_model_output (plms.py:178-190): guided eps, then the caller's score corrector
            e = guided(x, t_all[k], k)
            if score_corrector is not None:
                e = E._dev_f32(score_corrector.modify_score(self.model, e, x, t_long[k], conditioning, **(corrector_kwargs or {})), dev)
                guided.reclaim_context()
            return e
        inter = {"x_inter": [img], "pred_x0": [img]}
        old_eps = []
        paint = _Inpaint(self.model, mask, x0, size, kwargs.get("q_noise_fn"))
        for i in range(total):
            index = total - i - 1
            if paint.on:                     # plms.py:147-150
                img = paint.blend(img, steps[i])
            upd = lambda x, e: E.ddim_update(x, e, tb.alphas[index], tb.alphas_prev[index], 0.0,
                                             tb.sqrt_one_minus_alphas[index], None)
            e_t = eps_fn(img, i)
            if len(old_eps) == 0:          # pseudo improved Euler (plms.py:219-223)
                x_prev, _ = upd(img, e_t)
                e_next = eps_fn(x_prev, min(i + 1, total - 1))
                e_p = E.lincomb([(0.5, e_t), (0.5, e_next)])
            elif len(old_eps) == 1:
                e_p = E.lincomb([(1.5, e_t), (-0.5, old_eps[-1])])
            elif len(old_eps) == 2:
                e_p = E.lincomb([(23 / 12, e_t), (-16 / 12, old_eps[-1]), (5 / 12, old_eps[-2])])
            else:
                e_p = E.lincomb([(55 / 24, e_t), (-59 / 24, old_eps[-1]), (37 / 24, old_eps[-2]), (-9 / 24, old_eps[-3])])
            img, pred_x0 = upd(img, e_p)
            old_eps.append(e_t)
            if len(old_eps) >= 4:
                old_eps.pop(0)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total - 1:
                inter["x_inter"].append(img)
                inter["pred_x0"].append(pred_x0)
        return img, inter


class DPMSolverSampler(object):
    """DPM-Solver++(2M): multistep, order 2, time_uniform, predict_x0, lower_order_final (only if S < 15)."""

    def __init__(self, model, **kwargs):
        self.model = model

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, x_T=None, unconditional_guidance_scale=1.0,
               unconditional_conditioning=None, classifier=None, origin_cond=None, classifier_guide_scale=0.0,
               **kwargs):
        # the reference ignores eta / temperature here (sampler.py:24-56 passes neither to DPM_Solver): same
        reject_unsupported("DPMSolverSampler", kwargs, dict(mask=None, x0=None, score_corrector=None))   # sampler.py:24-56 accepts and drops them
        if S < 2:
            raise AssertionError("DPM-Solver++(2M) needs steps >= order = 2 (dpm_solver.py:1083 asserts steps >= order)")
        _warn_batch(conditioning, batch_size)
        dev = self.model.device
        C, H, W = shape
        size = (batch_size, C, H, W)
        x = torch.randn(size, device=dev) if x_T is None else x_T.to(dev, torch.float32).contiguous()
        ns = DPMTables(self.model.alphas_cumprod)
        ts = ns.time_steps(S)
        t_model = [ns.model_time(t) for t in ts]
        eps_fn = _Guided(self.model, conditioning, unconditional_guidance_scale, unconditional_conditioning, t_model[:S], size)
        t_in_all = torch.tensor(t_model, dtype=torch.float32, device=dev)[:, None].expand(S + 1, batch_size).contiguous()

        def model_fn(x, k):          # data prediction x0 = (x - sigma*eps)/alpha   (dpm_solver.py:386-393)
            t = ts[k]
            if classifier is None or not eps_fn.cfg:
                noise = eps_fn(x, t_in_all[k], k)
            else:                                          # double guidance (dpm_solver.py:1377-1393)
                noise, g = _eps_and_classifier_grad(self.model, classifier, lambda: eps_fn(x, t_in_all[k], k), x, t_in_all[k], origin_cond)
                noise = E.lincomb([(1.0, noise), (-classifier_guide_scale * ns.sigma(t), g)])
            a, s = ns.alpha(t), ns.sigma(t)
            return E.lincomb([(1.0 / a, x), (-s / a, noise)])

        def first_update(x, s, t, m_s):
            h = ns.lam(t) - ns.lam(s)
            return E.lincomb([(ns.sigma(t) / ns.sigma(s), x), (-ns.alpha(t) * np.expm1(-h), m_s)])

        def second_update(x, m1, m0, t1, t0, t):
            l1, l0, lt = ns.lam(t1), ns.lam(t0), ns.lam(t)
            h0, h = l0 - l1, lt - l0
            r0 = h0 / h
            k = ns.alpha(t) * (np.exp(-h) - np.float32(1.0))
            # x_t = sig_t/sig_0 x - k m0 - 0.5 k (m0 - m1)/r0
            return E.lincomb([(ns.sigma(t) / ns.sigma(t0), x), (-k - 0.5 * k / r0, m0), (0.5 * k / r0, m1)])

        m_prev = [model_fn(x, 0)]
        t_prev = [ts[0]]
        x = first_update(x, ts[0], ts[1], m_prev[-1])
        m_prev.append(model_fn(x, 1))
        t_prev.append(ts[1])
        for step in range(2, S + 1):
            order = min(2, S + 1 - step) if S < 15 else 2
            if order == 1:
                x = first_update(x, t_prev[-1], ts[step], m_prev[-1])
            else:
                x = second_update(x, m_prev[0], m_prev[1], t_prev[0], t_prev[1], ts[step])
            t_prev[0], m_prev[0] = t_prev[1], m_prev[1]
            t_prev[1] = ts[step]
            if step < S:
                m_prev[1] = model_fn(x, step)
        return x, None

    def sample_with_classifier(self, S, batch_size, shape, conditioning=None, origin_cond=None, classifier=None,
                               classifier_guide_scale=0.0, **kwargs):
        return self.sample(S, batch_size, shape, conditioning, origin_cond=origin_cond, classifier=classifier,
                           classifier_guide_scale=classifier_guide_scale, **kwargs)


@torch.no_grad()
def ancestral_sample(model, cond, shape, x_T=None, timesteps=None, log_every_t=None, return_intermediates=False,
                     noise_fn=None, callback=None, img_callback=None, mask=None, x0=None, q_noise_fn=None):
    """LatentDiffusion.p_sample_loop (ddpm.py:1201-1250): 1000-step ancestral sampling, no CFG, clip_denoised False."""
    dev = model.device
    img = torch.randn(shape, device=dev) if x_T is None else x_T.to(dev, torch.float32).contiguous()
    b = shape[0]
    T = model.num_timesteps if timesteps is None else timesteps
    log_every_t = log_every_t or model.log_every_t
    eps_fn = _Guided(model, cond, 1.0, None)
    tab = {k: getattr(model, k).cpu().numpy() for k in ("sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                                                        "posterior_mean_coef1", "posterior_mean_coef2",
                                                        "posterior_log_variance_clipped")}
    inter = [img]
    paint = _Inpaint(model, mask, x0, tuple(shape), q_noise_fn)
    for i in reversed(range(0, T)):
        ts = torch.full((b,), float(i), device=dev, dtype=torch.float32)
        eps = eps_fn(img, ts)
        # x_recon = c_r x - c_m eps ; mean = c1 x_recon + c2 x   (ddpm.py:221-234)
        cr, cm = tab["sqrt_recip_alphas_cumprod"][i], tab["sqrt_recipm1_alphas_cumprod"][i]
        c1, c2 = tab["posterior_mean_coef1"][i], tab["posterior_mean_coef2"][i]
        terms = [(c1 * cr + c2, img), (-c1 * cm, eps)]
        # noise is drawn at every step (also t == 0, where it is masked) like noise_like() in ddpm.py:1132
        noise = noise_fn(shape).to(dev) if noise_fn is not None else torch.randn(shape, device=dev)
        if i != 0:
            terms.append((np.exp(0.5 * tab["posterior_log_variance_clipped"][i]), noise))
        img = E.lincomb(terms)
        if paint.on:                         # ddpm.py:1239-1241: after the step
            img = paint.blend(img, i)
        if i % log_every_t == 0 or i == T - 1:
            inter.append(img)
        if callback:
            callback(i)
        if img_callback:
            img_callback(img, i)
    return (img, inter) if return_intermediates else img
