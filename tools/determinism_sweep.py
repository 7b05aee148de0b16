#!/usr/bin/env python
"""Run-to-run determinism of the LayerNorm-folded GEMM pair (df_test_ln_chain) for every tile that is valid as the consumer, at the
SpatialTransformer shapes of the B = 4 CFG step: the same call repeated REPS times must give the same bytes (t0, y, V^T).
usage: [DF_PRECISION=bf16|fp16] tools/determinism_sweep.py [reps]      (exit code 1 when any tile diverges)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_kernels_gpu as K
from test_kernels_gpu import rnd, bf, ptr, stream

K.PREC = os.environ.get("DF_PRECISION", "bf16")
E = K._eng(); L = E.lib(K.PREC); odt = K.odt()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
TILES = [int(x) for x in os.environ.get("TILES", "0,1,2,3,4,8,9,10,11,12,13,14,18,19,20,26,27,28,29").split(",")]
bad = []
for (M, C, T) in ((8192, 320, 1024), (2048, 640, 256), (512, 1280, 64)):
    A0 = bf(rnd((M, C), 40)).to(odt).cuda(); W0 = bf(rnd((C, C), 41) / C ** 0.5).to(odt).cuda()
    b0 = (0.1 * rnd((C,), 42)).cuda(); res = rnd((M, C), 43).cuda()
    gamma, beta = (1 + 0.2 * rnd((C,), 44)).cuda(), (0.2 * rnd((C,), 45)).cuda()
    for mode in (0, 1, 2):
        N1 = {0: C, 1: 8 * C if C <= 320 else 2 * C, 2: 3 * C}[mode]
        N1 = 8 * C if mode == 1 else N1
        W1 = (rnd((N1, C), 46) / C ** 0.5).cuda()
        b1 = None if mode == 2 else (0.1 * rnd((N1,), 47)).cuda()
        for tile in TILES:
            ref, nd, ok = None, 0, True
            for r in range(reps):
                t0 = torch.full((M, C), float("nan"), device="cuda")
                y = (torch.full((M, N1), float("nan"), device="cuda") if mode == 0 else
                     torch.full((M, N1 // 2 if mode == 1 else 2 * C), float("nan"), dtype=odt, device="cuda"))
                vt = torch.zeros(M // T, C, T, dtype=odt, device="cuda")
                rc = L.df_test_ln_chain(ptr(A0), ptr(W0), ptr(b0), ptr(res), ptr(gamma), ptr(beta), ptr(W1),
                                        ptr(b1) if b1 is not None else None, ptr(t0), ptr(y), ptr(vt), M, C, N1, mode, T, T,
                                        3, 1, tile, 1, stream())
                if rc != 0:
                    ok = False
                    break
                torch.cuda.synchronize()
                cur = (t0.view(torch.int32), y.view(torch.int32 if mode == 0 else torch.int16), vt.view(torch.int16))
                if ref is None:
                    ref = [c.clone() for c in cur]
                elif any(bool((a != b).any()) for a, b in zip(cur, ref)):
                    nd += 1
            if not ok:
                continue
            if nd:
                bad.append((M, C, mode, tile, nd))
            print(f"M {M} C {C} mode {mode} tile {tile}: {'DIVERGED %d/%d' % (nd, reps - 1) if nd else 'ok'}", flush=True)
print("diverging (M, C, mode, tile, count):", bad)
sys.exit(1 if bad else 0)
