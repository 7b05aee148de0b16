// Device side of experiments/aql/aql_fence_probe.cpp: one dependent "phase".  Block b reads the chunk block (b + 1 + phase % 3) % 256 -- a
// block of ANOTHER XCD -- wrote in the previous phase and writes its own.  No blockDim / gridDim (no hidden kernel arguments needed).
// build: hipcc --offload-arch=gfx950 -O3 --genco -o phase_kernel.hsaco phase_kernel.hip
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(256) void phase_k(const float* src, float* dst, int chunk, int phase) {
  const int b = __builtin_amdgcn_workgroup_id_x(), t = __builtin_amdgcn_workitem_id_x();
  const int other = (b + 1 + phase % 3) % 256;
  float acc = 0.f;
  for (int i = t; i < chunk; i += 256) acc += src[(size_t)other * chunk + i];
  for (int i = t; i < chunk; i += 256) dst[(size_t)b * chunk + i] = acc * 1e-3f + (float)(phase + 1);
}
// the same with write-through stores (sc1) -- what the engine's GEMM epilogues use
extern "C" __global__ __launch_bounds__(256) void phase_wt(const float* src, float* dst, int chunk, int phase) {
  const int b = __builtin_amdgcn_workgroup_id_x(), t = __builtin_amdgcn_workitem_id_x();
  const int other = (b + 1 + phase % 3) % 256;
  float acc = 0.f;
  for (int i = t; i < chunk; i += 256) acc += src[(size_t)other * chunk + i];
  for (int i = t; i < chunk; i += 256) {
    float v = acc * 1e-3f + (float)(phase + 1);
    float* p = dst + (size_t)b * chunk + i;
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  }
}
