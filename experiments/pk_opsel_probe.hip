// Probe (round 6): does a packed-FP32 VALU instruction whose LOW result lane takes the HIGH register of src1 (op_sel:[0,1]) compute
// correctly on MI355X when wavefronts of another workgroup share the SIMD?  Found in the fused-QKV GEMM's V^T epilogue (gemm_impl.h,
// `v[j].y += b.y` compiled to `v_pk_add_f32 v[8:9], v[8:9], v[28:29] op_sel:[0,1]`): with two blocks per CU, the low result of the last
// 16 lanes sometimes came out WITHOUT the addend; the commuted form `v_pk_add_f32 D, B, D op_sel:[1,0]` never did
// (tools/determinism_sweep.py).
//   every block runs two phases -- "noise" (MFMA + LDS + global traffic) and "victim" (the instruction under test on known data,
//   checked against scalar arithmetic on the spot) -- even blocks noise first, odd blocks victim first, LDS sized for 2 blocks per CU.
//   forms: 0 v_pk_add_f32 D, D, B op_sel:[0,1]     1 v_pk_add_f32 D, B, D op_sel:[1,0]     2 v_pk_mul_f32 D, D, B op_sel:[0,1]
//          3 v_pk_fma_f32 D, A, B, D op_sel:[0,1,0]  4 v_pk_fma_f32 D, D, ONE, B op_sel:[0,0,1] (addend = B.hi)
//          5 v_pk_add_f32 D, D, B op_sel_hi:[1,0] (HIGH result from src1's LOW register)   6 v_pk_mul_f32 D, D, B op_sel:[1,1]
//          7 v_pk_mov_b32 D, D, B op_sel:[0,1] (D.hi = B.hi)
// build: hipcc --offload-arch=gfx950 -O3 -o pk_opsel_probe pk_opsel_probe.hip ; run: ./pk_opsel_probe [lds_kb] [rounds] [other phase: 0 MFMA + LDS + global, 1 no MFMA, 2 none]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ float noise(float* lds, const float* g, int iters, float seed, int mode) {
  if (mode == 2) return 0.f;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = seed;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (float)(threadIdx.x & 15)); b[i] = (__bf16)(0.02f * (float)(i + 1)); }
  float s = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (mode == 0) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc, 0, 0, 0);
    }
    lds[(threadIdx.x * 33 + it) & 4095] = acc[it & 15];
    s += lds[(threadIdx.x * 17 + 5 * it) & 4095] + g[(blockIdx.x * 64 + threadIdx.x + it * 256) & 65535];
  }
  for (int i = 0; i < 16; ++i) s += acc[i];
  return s;
}

template <int FORM>
__device__ __forceinline__ void victim(const float4* bias, int iters, unsigned* bad, unsigned* lane_hist, float* sample) {
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
    const float4 b = bias[(blockIdx.x + it + (lane >> 5)) & 1023];      // half-wave uniform address, like the epilogue's bias read
    f32x2 d = {1.0f + 0.001f * (float)lane + (float)it, -2.0f + 0.003f * (float)lane};
    const f32x2 d0 = d;
    f32x2 B = {b.x, b.y};
    f32x2 A = {0.5f, 0.25f};
    const f32x2 ONE = {1.0f, 1.0f};
    float e0, e1;
    if (FORM == 0) { asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,1]" : "+v"(d) : "v"(B)); e0 = d0.x + b.y; e1 = d0.y + b.y; }
    if (FORM == 1) { asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[1,0] op_sel_hi:[1,1]" : "+v"(d) : "v"(B)); e0 = d0.x + b.y; e1 = d0.y + b.y; }
    if (FORM == 2) { asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,1]" : "+v"(d) : "v"(B)); e0 = d0.x * b.y; e1 = d0.y * b.y; }
    if (FORM == 3) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(d) : "v"(A), "v"(B)); e0 = __fmaf_rn(A.x, b.y, d0.x); e1 = __fmaf_rn(A.y, b.y, d0.y); }
    if (FORM == 4) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "+v"(d) : "v"(ONE), "v"(B)); e0 = __fmaf_rn(d0.x, 1.0f, b.y); e1 = __fmaf_rn(d0.y, 1.0f, b.y); }
    if (FORM == 5) { asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,0] op_sel_hi:[1,0]" : "+v"(d) : "v"(B)); e0 = d0.x + b.x; e1 = d0.y + b.x; }
    if (FORM == 6) { asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,1] op_sel_hi:[1,1]" : "+v"(d) : "v"(B)); e0 = d0.y * b.y; e1 = d0.y * b.y; }
    if (FORM == 7) { asm volatile("v_pk_mov_b32 %0, %0, %1 op_sel:[0,1]" : "+v"(d) : "v"(B)); e0 = d0.x; e1 = b.y; }
    if (d.x != e0 || d.y != e1) {
      if (atomicAdd(bad + (d.x != e0 ? 0 : 1), 1u) == 0u) {
        sample[0] = d.x; sample[1] = e0; sample[2] = d0.x; sample[3] = b.x; sample[4] = b.y; sample[5] = d.y; sample[6] = e1;
      }
      atomicAdd(lane_hist + lane, 1u);
    }
  }
}

template <int FORM>
__global__ __launch_bounds__(256) void probe(const float4* bias, const float* g, float* sink, unsigned* bad, unsigned* lane_hist, int iters, int mode, float* sample) {
  extern __shared__ float lds[];
  float s = 0.f;
  if (blockIdx.x & 1) {
    victim<FORM>(bias, iters, bad, lane_hist, sample);
    s = noise(lds, g, iters, 0.5f, mode);
  } else {
    s = noise(lds, g, iters, 0.25f, mode);
    victim<FORM>(bias, iters, bad, lane_hist, sample);
  }
  if (s == 123.456f) sink[0] = s;
}

template <int FORM>
void run(const float4* bias, const float* g, float* sink, unsigned* bad, unsigned* hist, int lds_bytes, int rounds, const char* name, int mode) {
  static float* sample = nullptr;
  if (!sample) CK(hipMalloc(&sample, 32));
  CK(hipMemset(sample, 0, 32));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  CK(hipMemset(bad, 0, 8));
  CK(hipMemset(hist, 0, 64 * 4));
  for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(probe<FORM>, dim3(1024), dim3(256), lds_bytes, 0, bias, g, sink, bad, hist, 400, mode, sample);
  CK(hipDeviceSynchronize());
  unsigned hb[2], hh[64];
  CK(hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hh, hist, 256, hipMemcpyDeviceToHost));
  unsigned q[4] = {0, 0, 0, 0};
  for (int i = 0; i < 64; ++i) q[i >> 4] += hh[i];
  float hs[8];
  CK(hipMemcpy(hs, sample, 32, hipMemcpyDeviceToHost));
  printf("form %d  %-40s wrong low %u, wrong high %u   by lane quarter [%u %u %u %u]", FORM, name, hb[0], hb[1], q[0], q[1], q[2], q[3]);
  if (hb[0]) printf("   e.g. low got %.6f expected %.6f (d.lo %.6f, B.lo %.6f, B.hi %.6f)", hs[0], hs[1], hs[2], hs[3], hs[4]);
  printf("\n");
}

int main(int argc, char** argv) {
  const int lds_kb = argc > 1 ? atoi(argv[1]) : 68, rounds = argc > 2 ? atoi(argv[2]) : 20, mode = argc > 3 ? atoi(argv[3]) : 0;
  std::vector<float> hbias(4096), hg(65536);
  for (size_t i = 0; i < hbias.size(); ++i) hbias[i] = 0.125f + 0.001f * (float)(i % 977);
  for (size_t i = 0; i < hg.size(); ++i) hg[i] = 0.001f * (float)(i % 31);
  float *bias, *g, *sink;
  unsigned *bad, *hist;
  CK(hipMalloc(&bias, hbias.size() * 4)); CK(hipMalloc(&g, hg.size() * 4)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&bad, 8)); CK(hipMalloc(&hist, 256));
  CK(hipMemcpy(bias, hbias.data(), hbias.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(g, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
  printf("1024 blocks x 256 threads, %d KB LDS per block (%d block(s) per CU), %d launches x 400 checks per lane, other phase: %s\n", lds_kb, 160 / lds_kb, rounds,
         mode == 0 ? "MFMA + LDS + global traffic" : mode == 1 ? "LDS + global traffic, no MFMA" : "none");
  run<0>((const float4*)bias, g, sink, bad, hist, lds_kb * 1024, rounds, "v_pk_add_f32 D, D, B op_sel:[0,1]", mode);
  run<1>((const float4*)bias, g, sink, bad, hist, lds_kb * 1024, rounds, "v_pk_add_f32 D, B, D op_sel:[1,0]", mode);
  run<2>((const float4*)bias, g, sink, bad, hist, lds_kb * 1024, rounds, "v_pk_mul_f32 D, D, B op_sel:[0,1]", mode);
  run<3>((const float4*)bias, g, sink, bad, hist, lds_kb * 1024, rounds, "v_pk_fma_f32 D, A, B, D op_sel:[0,1,0]", mode);
  run<4>((const float4*)bias, g, sink, bad, hist, lds_kb * 1024, rounds, "v_pk_fma_f32 D, D, 1, B op_sel:[0,0,1]", mode);
  run<5>((const float4*)bias, g, sink, bad, hist, lds_kb * 1024, rounds, "v_pk_add_f32 D, D, B op_sel_hi:[1,0]", mode);
  run<6>((const float4*)bias, g, sink, bad, hist, lds_kb * 1024, rounds, "v_pk_mul_f32 D, D, B op_sel:[1,1]", mode);
  run<7>((const float4*)bias, g, sink, bad, hist, lds_kb * 1024, rounds, "v_pk_mov_b32 D, D, B op_sel:[0,1]", mode);
  return 0;
}
