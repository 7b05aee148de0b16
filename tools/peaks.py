#!/usr/bin/env python
"""Measured device peaks on this box: MFMA bf16 issue rate, HBM streaming copy / read, and the
asymptotic rate of the engine's own GEMM kernel on a large square problem.  Writes profiles/peaks.json."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_foley_amd  # noqa
from diff_foley_amd import engine as E


def timeit(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    L = E.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {"device": torch.cuda.get_device_name(0), "cus": torch.cuda.get_device_properties(0).multi_processor_count}
    scratch = torch.zeros(64, device="cuda")
    best = 0
    for blocks in (256, 512, 1024, 2048):
        n = 20000
        t = timeit(lambda: L.df_test_peak(0, None, C.c_void_p(scratch.data_ptr()), n, blocks, st))
        tf = blocks * 4 * n * 4 * 2 * 32 * 32 * 16 / t / 1e12
        print(f"mfma bf16 32x32x16  blocks={blocks:5d}  {tf:8.1f} TFLOP/s")
        best = max(best, tf)
    out["mfma_bf16_tflops"] = round(best, 1)
    nbytes = 2 << 30
    src = torch.empty(nbytes // 4, device="cuda").normal_()
    dst = torch.empty_like(src)
    for kind, name in ((1, "copy"), (2, "read")):
        best = 0
        for blocks in (1024, 2048, 4096, 8192, 16384):
            t = timeit(lambda: L.df_test_peak(kind, C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), nbytes, blocks, st))
            gbs = nbytes * (2 if kind == 1 else 1) / t / 1e9
            print(f"hbm {name}  blocks={blocks:6d}  {gbs:8.1f} GB/s")
            best = max(best, gbs)
        out[f"hbm_{name}_gbs"] = round(best, 1)
    t = timeit(lambda: dst.copy_(src))
    out["hbm_torch_copy_gbs"] = round(2 * nbytes / t / 1e9, 1)
    print("torch copy_", out["hbm_torch_copy_gbs"], "GB/s")
    del src, dst
    # the engine's own GEMM kernel, large square problem (operands L2/MALL-friendly, K long): its asymptotic rate
    for (M, N, K) in ((8192, 8192, 8192), (16384, 4096, 4096)):
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
        c = torch.empty(M, N, device="cuda")
        for tile, name in ((0, "128x128"), (8, "128x256"), (9, "256x128")):
            p = lambda t: C.c_void_p(t.data_ptr())
            if L.df_test_gemm(p(a), p(w), p(c), M, N, K, tile, 1, st) != 0:
                continue
            t = timeit(lambda: L.df_test_gemm(p(a), p(w), p(c), M, N, K, tile, 1, st), 5)
            tf = 2.0 * M * N * K / t / 1e12
            print(f"engine gemm {M}x{N}x{K} tile {name}: {tf:7.1f} TFLOP/s")
            out[f"gemm_{M}x{N}x{K}_{name}_tflops"] = round(tf, 1)
        ref = timeit(lambda: torch.matmul(a, w.t()), 5)
        out[f"hipblaslt_{M}x{N}x{K}_tflops"] = round(2.0 * M * N * K / ref / 1e12, 1)
        print(f"torch.matmul (hipBLASLt) {M}x{N}x{K}: {out[f'hipblaslt_{M}x{N}x{K}_tflops']} TFLOP/s")
    os.makedirs("profiles", exist_ok=True)
    json.dump(out, open("profiles/peaks.json", "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
