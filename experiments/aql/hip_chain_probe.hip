// Companion of aql_fence_probe.cpp: the SAME chain of dependent phases submitted through HIP (stream launches, and one captured graph).
// build: hipcc --offload-arch=gfx950 -O3 -o hip_chain_probe hip_chain_probe.hip ; run: ./hip_chain_probe [chunk_floats]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void phase_wt(const float* src, float* dst, int chunk, int phase) {
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
int main(int argc, char** argv) {
  const int chunk = argc > 1 ? atoi(argv[1]) : 64;
  float *a, *b;
  CK(hipMalloc(&a, 256 * chunk * 4)); CK(hipMalloc(&b, 256 * chunk * 4));
  CK(hipMemset(a, 0, 256 * chunk * 4)); CK(hipMemset(b, 0, 256 * chunk * 4));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  auto chain = [&](int phases) {
    for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(phase_wt, dim3(256), dim3(256), 0, s, (p & 1) ? b : a, (p & 1) ? a : b, chunk, p);
  };
  auto timed = [&](int phases) {
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    chain(phases);
    CK(hipStreamSynchronize(s));
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  };
  for (int i = 0; i < 5; ++i) timed(240);
  // events around the chains (GPU-side time; the host issues ahead)
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto ev = [&](int phases) { CK(hipEventRecord(e0, s)); chain(phases); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return (double)ms * 1e3; };
  double t400 = 1e30, t2400 = 1e30;
  for (int r = 0; r < 10; ++r) { t400 = std::min(t400, ev(400)); t2400 = std::min(t2400, ev(2400)); }
  printf("HIP stream launches: %.2f us per phase (events; 400 phases %.1f us, 2400 phases %.1f us)\n", (t2400 - t400) / 2000.0, t400, t2400);
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  chain(240);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  auto gr = [&](int n) { CK(hipEventRecord(e0, s)); for (int i = 0; i < n; ++i) CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return (double)ms * 1e3; };
  gr(2);
  double g2 = 1e30, g10 = 1e30;
  for (int r = 0; r < 10; ++r) { g2 = std::min(g2, gr(2)); g10 = std::min(g10, gr(10)); }
  printf("HIP graph of 240 phases: %.2f us per phase (2 launches %.1f us, 10 launches %.1f us)\n", (g10 - g2) / (8 * 240.0), g2, g10);
  return 0;
}
