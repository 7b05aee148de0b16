#!/usr/bin/env python
"""Micro-benchmark of the MFMA implicit-GEMM kernel on the layer shapes of the UNet (N = 8) through the C ABI test
entry points.  Prints us and TFLOP/s for every (shape, tile, split-K).  Usage: python tools/gemm_bench.py [filter]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd  # noqa
from diff_foley_amd import engine as E

TILES = {0: "128x128", 1: "128x64", 2: "64x128", 3: "64x64", 4: "32x128", 5: "H128x64", 6: "H256x64", 7: "H128x128", 8: "128x256", 9: "256x128",
         10: "128x128s", 11: "128x64s", 12: "64x128s", 13: "64x64s", 14: "32x128s", 15: "H128x64d", 16: "H256x64d", 17: "H192x64",
         18: "P256x128", 19: "P128x128", 20: "P2_128x128", 23: "HP192x64", 24: "HP128x64", 25: "HP128x128", 26: "P64x64", 27: "P2_64x64", 28: "P128x64", 29: "P64x128"}
HALO = (5, 6, 7, 15, 16, 17, 23, 24, 25)
# (name, kind, NB, H, W, Cin, Cout) for conv ; (name, 'lin', M, N, K)
SHAPES = [
    ("conv 320->320 @16x64", "conv", 8, 16, 64, 320, 320),
    ("conv 640->320 @16x64", "conv", 8, 16, 64, 640, 320),
    ("conv 640->640 @8x32", "conv", 8, 8, 32, 640, 640),
    ("conv 1280->1280 @4x16", "conv", 8, 4, 16, 1280, 1280),
    ("conv 2560->1280 @2x8", "conv", 8, 2, 8, 2560, 1280),
    ("conv 1280->1280 @2x8", "conv", 8, 2, 8, 1280, 1280),
    # conv2 of the output-side ResBlocks with the 1x1 skip folded in as extra K: (name, 'skip', NB, H, W, Cin, Cout, Cin2)
    ("skip 320+960->320 @16x64", "skip", 8, 16, 64, 320, 320, 960),
    ("skip 320+640->320 @16x64", "skip", 8, 16, 64, 320, 320, 640),
    ("skip 640+1920->640 @8x32", "skip", 8, 8, 32, 640, 640, 1920),
    ("skip 640+960->640 @8x32", "skip", 8, 8, 32, 640, 640, 960),
    ("skip 1280+2560->1280 @4x16", "skip", 8, 4, 16, 1280, 1280, 2560),
    ("skip 1280+2560->1280 @2x8", "skip", 8, 2, 8, 1280, 1280, 2560),
    ("lin ff1 8192x2560x320", "lin", 8192, 2560, 320),
    ("lin ff2 8192x320x1280", "lin", 8192, 320, 1280),
    ("lin proj 8192x320x320", "lin", 8192, 320, 320),
    ("lin ff1 2048x5120x640", "lin", 2048, 5120, 640),
    ("lin proj 2048x640x640", "lin", 2048, 640, 640),
    ("lin ff1 512x10240x1280", "lin", 512, 10240, 1280),
    ("lin proj 512x1280x1280", "lin", 512, 1280, 1280),
]


def ptr(t):
    return C.c_void_p(t.data_ptr())


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3     # us


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    L = E.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for sh in SHAPES:
        if flt and flt not in sh[0]:
            continue
        if sh[1] == "conv":
            _, _, NB, H, W, Cin, Cout = sh
            a = torch.randn(NB * H * W, Cin, device="cuda").to(torch.bfloat16)
            w = (torch.randn(Cout, 9 * Cin, device="cuda") * 0.02).to(torch.bfloat16)
            b = torch.zeros(Cout, device="cuda")
            c = torch.empty(NB * H * W, Cout, device="cuda")
            flops = 2.0 * NB * H * W * Cout * 9 * Cin
            nk = 9 * Cin // 64
            call = lambda t, sk: L.df_test_conv3x3(ptr(a), ptr(w), ptr(b), ptr(c), NB, H, W, Cin, Cout, 1, 0, t, sk, st)
        elif sh[1] == "skip":
            _, _, NB, H, W, Cin, Cout, Cin2 = sh
            a = torch.randn(NB * H * W, Cin, device="cuda").to(torch.bfloat16)
            a2 = torch.randn(NB * H * W, Cin2, device="cuda").to(torch.bfloat16)
            w = (torch.randn(Cout, 9 * Cin + Cin2, device="cuda") * 0.02).to(torch.bfloat16)
            b = torch.zeros(Cout, device="cuda")
            c = torch.empty(NB * H * W, Cout, device="cuda")
            flops = 2.0 * NB * H * W * Cout * (9 * Cin + Cin2)
            nk = (9 * Cin + Cin2) // 64
            call = lambda t, sk: L.df_test_conv3x3_skip(ptr(a), ptr(a2), ptr(w), ptr(b), ptr(c), NB, H, W, Cin, Cin2, Cout, t, sk, st)
        else:
            _, _, M, N, K = sh
            a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
            w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
            c = torch.empty(M, N, device="cuda")
            flops = 2.0 * M * N * K
            nk = K // 64
            call = lambda t, sk: L.df_test_gemm(ptr(a), ptr(w), ptr(c), M, N, K, t, sk, st)
        res = []
        for t in TILES:
            if t in HALO and sh[1] not in ("conv", "skip"):
                continue
            for sk in (1, 2, 3, 4, 6, 8, 12, 16, 32):
                if sk > 1 and (nk // sk < 4 or (t in HALO and sh[5] // 64 // sk < 1)):
                    break
                if call(t, sk) != 0:
                    continue
                us = timeit(lambda: call(t, sk))
                res.append((us, t, sk))
        res.sort()
        top = len(res) if len(sys.argv) > 2 and sys.argv[2] == "all" else 5
        best = ", ".join(f"{TILES[t]}/sk{sk}: {us:6.1f}us {flops / us / 1e6:6.0f}TF" for us, t, sk in res[:top])
        print(f"{sh[0]:26s} {flops / 1e9:6.1f} GF | {best}")


if __name__ == "__main__":
    main()
