import sys, torch, traceback
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import diff_foley_amd as P
from diff_foley_amd import synth
from helpers import tiny_state_dict
m = P.LatentDiffusion(precision="fp16", **P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY))
m.load_state_dict(tiny_state_dict()); m.cuda()
def t(name, f):
    try:
        r = f()
        print(name, "->", tuple(r.shape) if hasattr(r, "shape") else r)
    except Exception as e:
        print(name, "RAISED", type(e).__name__, str(e)[:200])
t("cond B=0", lambda: m.get_learned_conditioning(torch.zeros(0, 32, 64).cuda()))
t("cond T=0", lambda: m.get_learned_conditioning(torch.zeros(2, 0, 64).cuda()))
t("decode B=0", lambda: m.decode_first_stage(torch.zeros(0, 4, 16, 64).cuda()))
c = m.get_learned_conditioning(synth.synthetic_cavp(2, 32, 64).cuda())
t("apply_model B=0", lambda: m.apply_model(torch.zeros(0, 4, 16, 64).cuda(), torch.zeros(0).cuda(), c[:0]))
t("sample B=0", lambda: m.sample_log_diff_sampler(c[:0], 0, "DDIM", 4)[0])
t("sample S=0", lambda: m.sample_log_diff_sampler(c, 2, "DDIM", 0)[0])
t("sample S=1", lambda: m.sample_log_diff_sampler(c, 2, "DDIM", 1)[0])
t("sample S=1000", lambda: m.sample_log_diff_sampler(c[:1], 1, "DDIM", 1000)[0])
t("sample S=2000", lambda: m.sample_log_diff_sampler(c[:1], 1, "DDIM", 2000)[0])
t("dpm S=1", lambda: m.sample_log_diff_sampler(c, 2, "DPM_Solver", 1)[0])
t("plms S=1", lambda: m.sample_log_diff_sampler(c, 2, "PLMS", 1)[0])
t("x non-contiguous", lambda: m.apply_model(torch.randn(2, 16, 64, 4).cuda().permute(0, 3, 1, 2), torch.tensor([5, 6]).cuda(), c))
t("x double", lambda: m.apply_model(torch.randn(2, 4, 16, 64).double().cuda(), torch.tensor([5, 6]).cuda(), c))
t("x cpu", lambda: m.apply_model(torch.randn(2, 4, 16, 64), torch.tensor([5, 6]), c))
t("cond batch mismatch", lambda: m.apply_model(torch.randn(2, 4, 16, 64).cuda(), torch.tensor([5, 6]).cuda(), c[:1]))
t("t scalar-like", lambda: m.apply_model(torch.randn(2, 4, 16, 64).cuda(), torch.tensor([5]).cuda(), c))
t("latent 15x64", lambda: m.apply_model(torch.randn(2, 4, 15, 64).cuda(), torch.tensor([5, 6]).cuda(), c))
t("nan input", lambda: m.apply_model(torch.full((2, 4, 16, 64), float('nan')).cuda(), torch.tensor([5, 6]).cuda(), c).isnan().any())
t("t=1e6", lambda: m.apply_model(torch.randn(2, 4, 16, 64).cuda(), torch.tensor([1e6, -5.0]).cuda(), c).isfinite().all())
from oracle import unet as ou
usd = ou.sub_state_dict(tiny_state_dict(), "model.diffusion_model.")
x = torch.randn(2, 4, 16, 64); tt = torch.tensor([1e6, -5.0])
print("t=1e6/-5 vs oracle", float((m.apply_model(x.cuda(), tt.cuda(), c).cpu() - ou.unet_forward(usd, synth.UNET_TINY, x, tt, c.cpu())).norm() / ou.unet_forward(usd, synth.UNET_TINY, x, tt, c.cpu()).norm()))
