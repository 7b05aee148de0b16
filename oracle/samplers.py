"""Oracle: sampler loops (DDIM, DPM-Solver++(2M), PLMS, ancestral DDPM).  Test infrastructure only.

Every function takes ``apply_model(x, t, c) -> eps`` (the oracle UNet or anything
with the same signature) and restates the host loop of the reference:

  * DDIMSampler.ddim_sampling / p_sample_ddim            diff_foley/models/diffusion/ddim.py:179-273
  * ... _with_classifier, cal_classifier_loglikelihood_grad   ddim.py:276-396, 333-341
  * DPMSolverSampler.sample[_with_classifier]             diff_foley/models/diffusion/dpm_solver/sampler.py:24-156
  * model_wrapper 'classifier-free' / 'double-guide'      dpm_solver.py:296-348, 1296-1393
  * DPM_Solver.sample (multistep, order 2, predict_x0)    dpm_solver.py:1071-1105
  * dpm_solver_first_update / multistep second update     dpm_solver.py:504-549, 755-810
  * PLMSSampler.plms_sampling / p_sample_plms             diff_foley/models/diffusion/plms.py:113-236
  * LatentDiffusion.p_sample_loop / p_sample / p_mean_variance   ddpm.py:1083-1250
  * inpainting (mask / x0): q_sample + blend               ddim.py:206-209, plms.py:147-150, ddpm.py:1239-1241, 279-282
"""
import numpy as np
import torch

from .schedule import NoiseScheduleVP, ddim_schedule


def _cfg_eps(apply_model, x, t, c, scale, uc):
    """CFG batch build + combine (ddim.py:237-245)."""
    if uc is None or scale == 1.0:
        return apply_model(x, t, c)
    x_in = torch.cat([x] * 2)
    t_in = torch.cat([t] * 2)
    c_in = torch.cat([uc, c])
    e_u, e_c = apply_model(x_in, t_in, c_in).chunk(2)
    return e_u + scale * (e_c - e_u)


def classifier_grad(classifier, x, t, origin_cond):
    """d/dx sum(log classifier(x,t,c))  (ddim.py:333-341 / dpm_solver.py:1340-1349), unscaled."""
    with torch.enable_grad():
        x_in = x.detach().requires_grad_(True)
        p = classifier(x_in, t, origin_cond)
        return torch.autograd.grad(torch.log(p).sum(), x_in)[0]


@torch.no_grad()
def q_sample_blend(img, x0, mask, alphas_cumprod, t, noise):
    """img_orig = q_sample(x0, t); img = img_orig * mask + (1 - mask) * img   (ddim.py:206-209 / plms.py:147-150 /
    ddpm.py:1239-1241; q_sample = sqrt(acp_t) x0 + sqrt(1 - acp_t) noise, ddpm.py:279-282)."""
    acp = torch.as_tensor(np.asarray(alphas_cumprod, dtype=np.float64))
    a = float(torch.sqrt(acp[int(t)]).to(torch.float32))
    b = float(torch.sqrt(1.0 - acp[int(t)]).to(torch.float32))
    img_orig = a * x0 + b * noise
    return img_orig * mask + (1.0 - mask) * img


def ddim_sample(apply_model, alphas_cumprod, S, x_T, cond, scale=1.0, uc=None, eta=0.0,
                log_every_t=100, classifier=None, origin_cond=None, classifier_scale=0.0,
                noise_fn=None, mask=None, x0=None, q_noise_fn=None, temperature=1.0, noise_dropout=0.0):
    sch = ddim_schedule(alphas_cumprod, S, eta)
    steps = sch["timesteps"]
    b = x_T.shape[0]
    img = x_T
    inter = {"x_inter": [img], "pred_x0": [img]}
    total = steps.shape[0]
    for i, step in enumerate(np.flip(steps)):
        index = total - i - 1
        ts = torch.full((b,), int(step), dtype=torch.long)
        a_t = torch.full((b, 1, 1, 1), float(sch["alphas"][index]))
        a_prev = torch.full((b, 1, 1, 1), float(sch["alphas_prev"][index]))
        sigma_t = torch.full((b, 1, 1, 1), float(sch["sigmas"][index]))
        s1m = torch.full((b, 1, 1, 1), float(sch["sqrt_one_minus_alphas"][index]))
        if mask is not None:                          # ddim.py:206-209
            qn = q_noise_fn(x0.shape) if q_noise_fn is not None else torch.randn_like(x0)
            img = q_sample_blend(img, x0, mask, alphas_cumprod, int(step), qn)
        e_t = _cfg_eps(apply_model, img, ts, cond, scale, uc)
        if classifier is not None:
            g = classifier_grad(classifier, img, ts, origin_cond) * classifier_scale
            e_t = e_t - (1 - a_t).sqrt() * g
        pred_x0 = (img - s1m * e_t) / a_t.sqrt()
        dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e_t
        # ddim.py:269-271: sigma_t * noise_like(...) * temperature, then F.dropout (functional default: training=True) on the noise
        noise = sigma_t * (noise_fn(img.shape) if noise_fn is not None else torch.randn(img.shape)) * temperature
        if noise_dropout > 0.0:
            noise = torch.nn.functional.dropout(noise, p=noise_dropout)
        img = a_prev.sqrt() * pred_x0 + dir_xt + noise
        if index % log_every_t == 0 or index == total - 1:
            inter["x_inter"].append(img)
            inter["pred_x0"].append(pred_x0)
    return img, inter


@torch.no_grad()
def dpm_solver_sample(apply_model, alphas_cumprod, S, x_T, cond, scale=1.0, uc=None,
                      classifier=None, origin_cond=None, classifier_scale=0.0):
    """DPM-Solver++(2M), time_uniform, multistep order 2, lower_order_final (only if S<15)."""
    ns = NoiseScheduleVP(alphas_cumprod)
    b = x_T.shape[0]

    def model_fn(x, t_cont):                       # data-prediction wrapper (dpm_solver.py:386-393)
        t_in = (t_cont - 1.0 / ns.total_N) * 1000.0
        if uc is None or scale == 1.0:
            noise = apply_model(x, t_in, cond)
        else:
            noise = _cfg_eps(apply_model, x, t_in, cond, scale, uc)
            if classifier is not None:            # double-guide (dpm_solver.py:1377-1393)
                g = classifier_grad(classifier, x, t_in, origin_cond)
                sigma = ns.marginal_std(t_cont)
                noise = noise - classifier_scale * sigma.reshape(-1, 1, 1, 1) * g
        a, s = ns.marginal_alpha(t_cont), ns.marginal_std(t_cont)
        return (x - s.reshape(-1, 1, 1, 1) * noise) / a.reshape(-1, 1, 1, 1)

    e4 = lambda v: v.reshape(-1, 1, 1, 1)

    def first_update(x, s, t, model_s):
        lam_s, lam_t = ns.marginal_lambda(s), ns.marginal_lambda(t)
        h = lam_t - lam_s
        sig_s, sig_t = ns.marginal_std(s), ns.marginal_std(t)
        alpha_t = torch.exp(ns.marginal_log_mean_coeff(t))
        return e4(sig_t / sig_s) * x - e4(alpha_t * torch.expm1(-h)) * model_s

    def second_update(x, models, ts, t):
        m1, m0 = models
        t1, t0 = ts
        l1, l0, lt = ns.marginal_lambda(t1), ns.marginal_lambda(t0), ns.marginal_lambda(t)
        sig0, sigt = ns.marginal_std(t0), ns.marginal_std(t)
        alpha_t = torch.exp(ns.marginal_log_mean_coeff(t))
        h0, h = l0 - l1, lt - l0
        r0 = h0 / h
        D1 = e4(1.0 / r0) * (m0 - m1)
        return (e4(sigt / sig0) * x - e4(alpha_t * (torch.exp(-h) - 1.0)) * m0
                - 0.5 * e4(alpha_t * (torch.exp(-h) - 1.0)) * D1)

    timesteps = torch.linspace(1.0, 1.0 / ns.total_N, S + 1)
    x = x_T
    vec_t = timesteps[0].expand(b)
    models, tl = [model_fn(x, vec_t)], [vec_t]
    vec_t = timesteps[1].expand(b)
    x = first_update(x, tl[-1], vec_t, models[-1])
    models.append(model_fn(x, vec_t))
    tl.append(vec_t)
    for step in range(2, S + 1):
        vec_t = timesteps[step].expand(b)
        order = min(2, S + 1 - step) if S < 15 else 2
        if order == 1:
            x = first_update(x, tl[-1], vec_t, models[-1])
        else:
            x = second_update(x, models, tl, vec_t)
        tl[0], models[0] = tl[1], models[1]
        tl[1] = vec_t
        if step < S:
            models[1] = model_fn(x, vec_t)
    return x, None


@torch.no_grad()
def plms_sample(apply_model, alphas_cumprod, S, x_T, cond, scale=1.0, uc=None, log_every_t=100, mask=None, x0=None,
                q_noise_fn=None):
    sch = ddim_schedule(alphas_cumprod, S, 0.0)
    steps = sch["timesteps"]
    b = x_T.shape[0]
    img = x_T
    inter = {"x_inter": [img], "pred_x0": [img]}
    time_range = np.flip(steps)
    total = steps.shape[0]
    old_eps = []

    def x_prev_pred(x, e, index):
        a_t = float(sch["alphas"][index])
        a_prev = float(sch["alphas_prev"][index])
        s1m = float(sch["sqrt_one_minus_alphas"][index])
        a_t, a_prev, s1m = (torch.full((b, 1, 1, 1), v) for v in (a_t, a_prev, s1m))
        pred_x0 = (x - s1m * e) / a_t.sqrt()
        return a_prev.sqrt() * pred_x0 + (1.0 - a_prev).sqrt() * e, pred_x0

    for i, step in enumerate(time_range):
        index = total - i - 1
        ts = torch.full((b,), int(step), dtype=torch.long)
        ts_next = torch.full((b,), int(time_range[min(i + 1, len(time_range) - 1)]), dtype=torch.long)
        if mask is not None:                          # plms.py:147-150
            qn = q_noise_fn(x0.shape) if q_noise_fn is not None else torch.randn_like(x0)
            img = q_sample_blend(img, x0, mask, alphas_cumprod, int(step), qn)
        e_t = _cfg_eps(apply_model, img, ts, cond, scale, uc)
        if len(old_eps) == 0:
            x_prev, _ = x_prev_pred(img, e_t, index)
            e_next = _cfg_eps(apply_model, x_prev, ts_next, cond, scale, uc)
            e_p = (e_t + e_next) / 2
        elif len(old_eps) == 1:
            e_p = (3 * e_t - old_eps[-1]) / 2
        elif len(old_eps) == 2:
            e_p = (23 * e_t - 16 * old_eps[-1] + 5 * old_eps[-2]) / 12
        else:
            e_p = (55 * e_t - 59 * old_eps[-1] + 37 * old_eps[-2] - 9 * old_eps[-3]) / 24
        img, pred_x0 = x_prev_pred(img, e_p, index)
        old_eps.append(e_t)
        if len(old_eps) >= 4:
            old_eps.pop(0)
        if index % log_every_t == 0 or index == total - 1:
            inter["x_inter"].append(img)
            inter["pred_x0"].append(pred_x0)
    return img, inter


@torch.no_grad()
def ddpm_sample(apply_model, sched, x_T, cond, timesteps=None, noise_fn=None, log_every_t=200, mask=None, x0=None,
                q_noise_fn=None):
    """Ancestral sampling, no CFG (ddpm.py:1201-1250, p_sample :1115-1143, p_mean_variance :1083-1112,
    clip_denoised False for LatentDiffusion, ddpm.py:475)."""
    T = sched["betas"].shape[0] if timesteps is None else timesteps
    b = x_T.shape[0]
    img = x_T
    inter = [img]
    ex = lambda a, t: a.gather(-1, t).reshape(b, 1, 1, 1)
    for i in reversed(range(0, T)):
        ts = torch.full((b,), i, dtype=torch.long)
        eps = apply_model(img, ts, cond)
        x_recon = ex(sched["sqrt_recip_alphas_cumprod"], ts) * img - ex(sched["sqrt_recipm1_alphas_cumprod"], ts) * eps
        mean = ex(sched["posterior_mean_coef1"], ts) * x_recon + ex(sched["posterior_mean_coef2"], ts) * img
        logvar = ex(sched["posterior_log_variance_clipped"], ts)
        noise = noise_fn(img.shape) if noise_fn is not None else torch.randn(img.shape)
        nonzero = (1 - (ts == 0).float()).reshape(b, 1, 1, 1)
        img = mean + nonzero * (0.5 * logvar).exp() * noise
        if mask is not None:                          # ddpm.py:1239-1241 (after the step)
            qn = q_noise_fn(x0.shape) if q_noise_fn is not None else torch.randn_like(x0)
            img = (ex(sched["sqrt_alphas_cumprod"], ts) * x0 + ex(sched["sqrt_one_minus_alphas_cumprod"], ts) * qn) * mask \
                + (1.0 - mask) * img
        if i % log_every_t == 0 or i == T - 1:
            inter.append(img)
    return img, inter
