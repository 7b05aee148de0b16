// Device-peak microbenchmarks ("peak denominators must be replaced by numbers measured on the
// box").  Not on the product path; tools/peaks.py times them and DESIGN.md quotes the results next to the nominal
// peaks that bench.py's roofline uses.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/df_engine.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// 4 independent accumulator chains per wave so the MFMA pipe never waits on a dependent result.
__global__ __launch_bounds__(256) void peak_mfma_kernel(float* out, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 7); b[i] = (__bf16)1.0f; }
  f32x16_t c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (s == 12345.678f) out[0] = s;   // keep the chain alive without a store on the timed path
#endif
}

__global__ __launch_bounds__(256) void peak_copy_kernel(const f32x4_t* __restrict__ src, f32x4_t* __restrict__ dst, size_t n4) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    f32x4_t v0 = __builtin_nontemporal_load(src + i), v1 = __builtin_nontemporal_load(src + i + stride);
    f32x4_t v2 = __builtin_nontemporal_load(src + i + 2 * stride), v3 = __builtin_nontemporal_load(src + i + 3 * stride);
    __builtin_nontemporal_store(v0, dst + i); __builtin_nontemporal_store(v1, dst + i + stride);
    __builtin_nontemporal_store(v2, dst + i + 2 * stride); __builtin_nontemporal_store(v3, dst + i + 3 * stride);
  }
  for (; i < n4; i += stride) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void peak_read_kernel(const f32x4_t* __restrict__ src, float* out, size_t n4) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  f32x4_t acc = {0, 0, 0, 0};
  for (; i + 3 * stride < n4; i += 4 * stride) {
    f32x4_t v0 = __builtin_nontemporal_load(src + i), v1 = __builtin_nontemporal_load(src + i + stride);
    f32x4_t v2 = __builtin_nontemporal_load(src + i + 2 * stride), v3 = __builtin_nontemporal_load(src + i + 3 * stride);
    acc += v0 + v1 + v2 + v3;
  }
  for (; i < n4; i += stride) acc += src[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

extern "C" {
// kind 0: MFMA issue peak, `n` = iterations per wave (4 MFMA 32x32x16 bf16 each), grid = blocks x 4 waves.
//         FLOPs per launch = blocks * 4 waves * n * 4 * 2*32*32*16.
// kind 1: streaming copy of n bytes src -> dst (HBM bytes moved = 2n).   kind 2: streaming read of n bytes.
int df_test_peak(int kind, const void* src, void* dst, size_t n, int blocks, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (kind == 0) peak_mfma_kernel<<<blocks, 256, 0, st>>>((float*)dst, (int)n);
  else if (kind == 1) peak_copy_kernel<<<blocks, 256, 0, st>>>((const f32x4_t*)src, (f32x4_t*)dst, n / 16);
  else if (kind == 2) peak_read_kernel<<<blocks, 256, 0, st>>>((const f32x4_t*)src, (float*)dst, n / 16);
  else return 1;
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
}
