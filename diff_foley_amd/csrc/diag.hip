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

// L2 -> CU fill-rate probes: every block re-reads a `span`-byte window (L2-resident when the windows of all blocks
// fit the 4 MB/XCD L2) `n` times.  kind 3: LDS-DMA (buffer_load_dwordx4 ... lds);  kind 4: global_load_dwordx4 into
// VGPRs (values summed);  kind 5: global_load_dwordx4 + ds_write_b128.
template <int KIND>
__global__ __launch_bounds__(256) void fill_probe_kernel(const char* __restrict__ src, float* out, size_t span, int n,
                                                         size_t stride_blk) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 64 KB
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const char* base = src + (size_t)blockIdx.x * stride_blk;
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)span, 0x00020000);
  f32x4_t acc = {0, 0, 0, 0};
  const int chunks = (int)(span / 16384);            // 16 KB per step of the block (4 x 4 KB passes)
  for (int it = 0; it < n; ++it) {
    for (int c = 0; c < chunks; ++c) {
      const unsigned off = (unsigned)c * 16384u + (unsigned)tid * 16u;
      if (KIND == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + ((c & 3) * 16384 + i * 4096 + wid * 1024)), 16,
                                                   off + i * 4096, 0, 0, 0);
      } else {
        f32x4_t v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4_t*>(base + off + i * 4096);
        if (KIND == 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) acc += v[i];
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            *reinterpret_cast<f32x4_t*>(smem + (c & 3) * 16384 + i * 4096 + tid * 16) = v[i];
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (KIND != 4) acc[0] = *reinterpret_cast<float*>(smem + tid * 4);
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
#endif
}

extern "C" {
// kind 0: MFMA issue peak, `n` = iterations per wave (4 MFMA 32x32x16 bf16 each), grid = blocks x 4 waves.
//         FLOPs per launch = blocks * 4 waves * n * 4 * 2*32*32*16.
// kind 1: streaming copy of n bytes src -> dst (HBM bytes moved = 2n).   kind 2: streaming read of n bytes.
// L2 -> CU fill probes (kinds 3..5, see fill_probe_kernel): span bytes per block window, stride_blk bytes between the
// windows of consecutive blocks (0 = all blocks share one window), n repetitions.
int df_test_fill(int kind, const void* src, void* dst, size_t span, size_t stride_blk, int n, int blocks, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_probe_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_probe_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_probe_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    attr = true;
  }
  if (kind == 3) fill_probe_kernel<3><<<blocks, 256, 65536, st>>>((const char*)src, (float*)dst, span, n, stride_blk);
  else if (kind == 4) fill_probe_kernel<4><<<blocks, 256, 65536, st>>>((const char*)src, (float*)dst, span, n, stride_blk);
  else if (kind == 5) fill_probe_kernel<5><<<blocks, 256, 65536, st>>>((const char*)src, (float*)dst, span, n, stride_blk);
  else return 1;
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

int df_test_peak(int kind, const void* src, void* dst, size_t n, int blocks, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (kind == 0) peak_mfma_kernel<<<blocks, 256, 0, st>>>((float*)dst, (int)n);
  else if (kind == 1) peak_copy_kernel<<<blocks, 256, 0, st>>>((const f32x4_t*)src, (f32x4_t*)dst, n / 16);
  else if (kind == 2) peak_read_kernel<<<blocks, 256, 0, st>>>((const f32x4_t*)src, (float*)dst, n / 16);
  else return 1;
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
}
