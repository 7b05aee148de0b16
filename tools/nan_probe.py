"""Which op first stores a non-finite operand value?  Full UNet, bf16 build, B = 4 CFG forward(s) (df_debug_saturations)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd as P
from diff_foley_amd import synth
m = P.LatentDiffusion(precision="bf16", **P.stage2_config())
m.load_state_dict(synth.make_state_dict(synth.state_dict_spec(), 0))
m.cuda()
if os.environ.get("NAN_TUNE"): m.autotune(True)
xT = synth.synthetic_xT(4, seed=21).cuda()
c = m.get_learned_conditioning(synth.synthetic_cavp(4, 32, 512, seed=1234).cuda())
uc = torch.zeros_like(c)
eng = m.engine
eng.set_context(torch.cat([uc, c]))
eng.unet_forward_cfg(xT, torch.full((4,), 961.0, device="cuda"), 4.5)      # builds (and, with NAN_TUNE, tunes) the plan before counting
torch.cuda.synchronize()
eng.debug_saturations(True)
x = xT
for step, t in enumerate([961.0, 761.0, 561.0]):
    y = eng.unet_forward_cfg(x, torch.full((4,), t, device="cuda"), 4.5)
    res = eng.debug_saturations_read()
    bad = [(i, lab, n) for i, (lab, n) in enumerate(res) if n]
    print("step", step, "finite out:", bool(torch.isfinite(y).all()), "ops:", len(res), "bad:", bad[:6])
    if bad:
        i0 = bad[0][0]
        print("ops in front of the first flagged one:", [res[k][0] for k in range(max(0, i0 - 3), i0 + 1)])
        eng.debug_saturations(False)
        eng.profile_begin(); eng.unet_forward_cfg(xT, torch.full((4,), 961.0, device="cuda"), 4.5); torch.cuda.synchronize(); eng.profile_end()
        eng.profile_dump("/tmp/nan_ops.csv")
        rows = open("/tmp/nan_ops.csv").read().splitlines()
        tag = bad[0][1].split(":")[1]; idx = int(bad[0][1].split("#")[1].split(":")[0])
        print("plan rows around it:", rows[idx - 1:idx + 2])
        break
    x = x - 0.1 * y
