// Probe: do two chains of dependent small kernels on two HIP streams of ONE process overlap on MI355X?
// (Two processes sharing the GPU do: 2 x bench.py at N=4 finish a step pair in 4.9 ms vs 7.2 ms back to back.)
// build: hipcc --offload-arch=gfx950 -O3 -o stream_overlap_probe stream_overlap_probe.hip ; run: ./stream_overlap_probe [blocks] [lds_kb] [us]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void busy(float* p, long cycles) {
  extern __shared__ float sm[];
  sm[threadIdx.x] = p[blockIdx.x];
  __syncthreads();
  const long t0 = (long)wall_clock64();                  // 100 MHz constant counter
  while ((long)wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(4);
  if (sm[threadIdx.x] == 123.f) p[blockIdx.x] = 1.f;
}

int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 160;
  const int lds = (argc > 2 ? atoi(argv[2]) : 0) * 1024;
  const long cyc = (argc > 3 ? atol(argv[3]) : 5) * 100;   // us -> 100 MHz ticks
  const int L = 300;
  const bool solo = argc > 4;          // "solo": one chain only, long enough to overlap with a second PROCESS doing the same
  float* p;
  CK(hipMalloc(&p, 4096 * 4));
  CK(hipMemset(p, 0, 4096 * 4));
  if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*)busy, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipStream_t s[2];
  CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  auto chain = [&](hipStream_t st) { for (int i = 0; i < L; ++i) hipLaunchKernelGGL(busy, dim3(nb), dim3(256), lds > 1024 ? lds : 1024, st, p, cyc); };
  for (int rep = 0; rep < 2; ++rep) { chain(s[0]); chain(s[1]); }
  CK(hipDeviceSynchronize());
  if (solo) {
    CK(hipEventRecord(e0, s[0]));
    for (int r = 0; r < 20; ++r) chain(s[0]);
    CK(hipEventRecord(e1, s[0]));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("solo process: blocks %d lds %d KB busy %ld us: %.2f us/kernel\n", nb, lds / 1024, cyc / 100, ms * 1000 / (20 * L));
    return 0;
  }
  // one chain
  CK(hipEventRecord(e0, s[0]));
  chain(s[0]);
  CK(hipEventRecord(e1, s[0]));
  CK(hipEventSynchronize(e1));
  float one; CK(hipEventElapsedTime(&one, e0, e1));
  // two chains on two streams, issued alternately kernel by kernel
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, s[0]));
  CK(hipStreamWaitEvent(s[1], e0, 0));
  for (int i = 0; i < L; ++i) {
    hipLaunchKernelGGL(busy, dim3(nb), dim3(256), lds > 1024 ? lds : 1024, s[0], p, cyc);
    hipLaunchKernelGGL(busy, dim3(nb), dim3(256), lds > 1024 ? lds : 1024, s[1], p, cyc);
  }
  CK(hipEventRecord(e1, s[0]));
  CK(hipEventRecord(e2, s[1]));
  CK(hipEventSynchronize(e1)); CK(hipEventSynchronize(e2));
  float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e0, e2));
  const float two = a > b ? a : b;
  printf("blocks %d lds %d KB busy %ld us: one chain %.2f us/kernel, two chains %.2f us per kernel pair -> overlap x%.2f\n", nb, lds / 1024,
         cyc / 100, one * 1000 / L, two * 1000 / L, 2 * one / two);
  return 0;
}
