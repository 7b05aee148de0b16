#!/usr/bin/env python
"""Debug helper for ffn_wide.hip: df_test_geglu on one shape, per-block error map against fp32."""
import ctypes as C, os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_foley_amd import engine as E
prec = os.environ.get("WPREC", "bf16")
L = E.lib(prec); odt = E.OPERAND_DTYPE[prec]
ptr = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
tile, M, Cc = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
N1 = 8 * Cc
g = torch.Generator().manual_seed(1)
x = torch.randn(M, Cc, generator=g) * 1.5 + 0.3
W = torch.randn(N1, Cc, generator=g) * 0.06
bias = torch.randn(N1, generator=g) * 0.2
cs = W.to(odt).float().sum(1)
xs = x.view(M, Cc // 64, 64)
stats = torch.stack([xs.sum(2), (xs * xs).sum(2)], dim=-1).contiguous()
A = x.to(odt).float()
mean = x.mean(1, keepdim=True); rstd = torch.rsqrt((x * x).mean(1, keepdim=True) - mean * mean + 1e-5)
v = (rstd * (A @ W.to(odt).float().t() - mean * cs[None]) + bias[None]).view(M, -1, 2, 32)
ref = (v[:, :, 0] * F.gelu(v[:, :, 1])).reshape(M, -1)
out = torch.full((M, N1 // 2), float("nan"), device="cuda", dtype=odt)
ad, wd, sd, cd, bd = x.to(odt).cuda(), W.to(odt).cuda(), stats.cuda(), cs.cuda(), bias.cuda()
rc = L.df_test_geglu(ptr(ad), ptr(wd), ptr(sd), ptr(cd), ptr(bd), ptr(out), M, Cc, N1, tile, int(os.environ.get("WDBG", "0")), st)
torch.cuda.synchronize()
print("rc", rc, L.df_last_error() if rc else "")
got = out.float().cpu()
fin = torch.isfinite(got)
print("finite fraction", float(fin.float().mean()))
RB, CB = 64, 80
for r0 in range(0, M, RB):
    line = []
    for c0 in range(0, N1 // 2, CB):
        gb, rb = got[r0:r0 + RB, c0:c0 + CB], ref[r0:r0 + RB, c0:c0 + CB]
        f = torch.isfinite(gb)
        if not f.all():
            line.append("NaN%3d" % int((~f).float().mean() * 100))
        else:
            line.append("%6.3f" % float((gb - rb).norm() / rb.norm()))
    print("rows %4d:" % r0, " ".join(line[:24]))
