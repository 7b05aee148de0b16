// Probe: what does a grid-wide barrier inside ONE persistent kernel cost on MI355X, against a kernel boundary?
// (Design question for a persistent SpatialTransformer chain kernel: DESIGN.md section 5.)
//   arm 0: R dependent launches of a kernel in which every block writes CHUNK bytes and reads another block's chunk
//   arm 1: one cooperative launch, R phases separated by an atomic-counter grid barrier (release/acquire, agent scope)
//   arm 2: as arm 1 but with a 16 KB "weight" read per phase issued BEFORE the barrier (what a chain kernel would do)
// build: hipcc --offload-arch=gfx950 -O3 -o gridbar_probe gridbar_probe.hip ; run: ./gridbar_probe [blocks] [chunk_floats]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __atomic_thread_fence(__ATOMIC_RELEASE);   // agent scope on a device pointer: L2 write-back
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  __syncthreads();
}

__device__ __forceinline__ float phase_body(const float* src, float* dst, int chunk, int phase, int nb) {
  // read the chunk the "previous phase" of ANOTHER block wrote (different XCD: +1 block = next XCD), write mine
  const int other = (blockIdx.x + 1 + phase) % nb;
  float acc = 0.f;
  for (int i = threadIdx.x; i < chunk; i += blockDim.x) acc += src[(size_t)other * chunk + i];
  for (int i = threadIdx.x; i < chunk; i += blockDim.x) dst[(size_t)blockIdx.x * chunk + i] = acc * 1e-3f + (float)(phase + 1);
  return acc;
}

// arm 3/4: no bulk L2 write-back / invalidate: the exchanged data moves with agent-scope (sc1) loads and stores, the barrier is
// a relaxed counter + s_waitcnt
__device__ __forceinline__ void grid_barrier_nofence(unsigned* counter, unsigned target) {
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}
// arm 4: hierarchical arrival (one counter per XCD, blocks round-robin over the 8 XCDs, the last arriver of an XCD bumps the
// global counter) and a slower poll
__device__ __forceinline__ void grid_barrier_hier(unsigned* counters, unsigned phase, unsigned nb) {
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned x = blockIdx.x & 7, per = (nb + 7 - x) / 8;
    const unsigned old = __hip_atomic_fetch_add(counters + 32 * (1 + x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == phase * per) __hip_atomic_fetch_add(counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase * 8) __builtin_amdgcn_s_sleep(8);
  }
  __syncthreads();
}
__global__ void persistent_hier(float* a, float* b, int chunk, int phases, int nb, unsigned* counters, float* sink);
__device__ __forceinline__ float phase_body_sc1(const float* src, float* dst, int chunk, int phase, int nb) {
  const int other = (blockIdx.x + 1 + phase) % nb;
  float acc = 0.f;
  for (int i = threadIdx.x; i < chunk; i += blockDim.x)
    acc += __hip_atomic_load(src + (size_t)other * chunk + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int i = threadIdx.x; i < chunk; i += blockDim.x)
    __hip_atomic_store(dst + (size_t)blockIdx.x * chunk + i, acc * 1e-3f + (float)(phase + 1), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  return acc;
}
__global__ void persistent_sc1(float* a, float* b, int chunk, int phases, int nb, unsigned* counter, float* sink) {
  float keep = 0.f;
  for (int p = 0; p < phases; ++p) {
    const float* src = (p & 1) ? b : a;
    float* dst = (p & 1) ? a : b;
    if (p) grid_barrier_nofence(counter, (unsigned)(p * nb));
    keep += phase_body_sc1(src, dst, chunk, p, nb);
  }
  if (keep == 123.456f) sink[0] = keep;
}

__global__ void persistent_hier(float* a, float* b, int chunk, int phases, int nb, unsigned* counters, float* sink) {
  float keep = 0.f;
  for (int p = 0; p < phases; ++p) {
    const float* src = (p & 1) ? b : a;
    float* dst = (p & 1) ? a : b;
    if (p) grid_barrier_hier(counters, (unsigned)p, (unsigned)nb);
    keep += phase_body_sc1(src, dst, chunk, p, nb);
  }
  if (keep == 123.456f) sink[0] = keep;
}

__global__ void one_phase(const float* src, float* dst, int chunk, int phase, int nb) { phase_body(src, dst, chunk, phase, nb); }

__global__ void persistent(float* a, float* b, int chunk, int phases, int nb, unsigned* counter, const float* w, int prefetch,
                           float* sink) {
  __shared__ float wl[4096];
  float keep = 0.f;
  for (int p = 0; p < phases; ++p) {
    const float* src = (p & 1) ? b : a;
    float* dst = (p & 1) ? a : b;
    if (prefetch)
      for (int i = threadIdx.x; i < 4096; i += blockDim.x) wl[i] = w[((size_t)p * nb + blockIdx.x) % 64 * 4096 + i];
    if (p) grid_barrier(counter, (unsigned)(p * nb));
    keep += phase_body(src, dst, chunk, p, nb);
    if (prefetch) keep += wl[threadIdx.x];
  }
  if (keep == 123.456f) sink[0] = keep;
}

int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 256;
  const int chunk = argc > 2 ? atoi(argv[2]) : 4096;
  const int R = 40, reps = 20;
  float *a, *b, *w, *sink;
  unsigned* counter;
  CK(hipMalloc(&a, (size_t)nb * chunk * 4));
  CK(hipMalloc(&b, (size_t)nb * chunk * 4));
  CK(hipMalloc(&w, 64 * 4096 * 4));
  CK(hipMalloc(&sink, 4));
  CK(hipMalloc(&counter, 4 * 32 * 9));
  CK(hipMemset(a, 0, (size_t)nb * chunk * 4));
  CK(hipMemset(b, 0, (size_t)nb * chunk * 4));
  CK(hipMemset(w, 0, 64 * 4096 * 4));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  int maxb = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&maxb, persistent, 256, 0));
  printf("blocks %d chunk %d floats (%.1f KB/block), occupancy %d blocks/CU\n", nb, chunk, chunk * 4 / 1024.0, maxb);
  std::vector<float> h0((size_t)nb * chunk), h1((size_t)nb * chunk);
  for (int arm = 0; arm < 5; ++arm) {
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
      CK(hipMemsetAsync(counter, 0, 4 * 32 * 9, s));
      CK(hipMemsetAsync(a, 0, (size_t)nb * chunk * 4, s));
      CK(hipEventRecord(e0, s));
      if (arm == 0) {
        for (int p = 0; p < R; ++p)
          hipLaunchKernelGGL(one_phase, dim3(nb), dim3(256), 0, s, (p & 1) ? b : a, (p & 1) ? a : b, chunk, p, nb);
      } else if (arm == 4) {
        int phases = R;
        void* args[] = {&a, &b, (void*)&chunk, &phases, (void*)&nb, &counter, &sink};
        CK(hipLaunchCooperativeKernel((const void*)persistent_hier, dim3(nb), dim3(256), args, 0, s));
      } else if (arm == 3) {
        int phases = R;
        void* args[] = {&a, &b, (void*)&chunk, &phases, (void*)&nb, &counter, &sink};
        CK(hipLaunchCooperativeKernel((const void*)persistent_sc1, dim3(nb), dim3(256), args, 0, s));
      } else {
        int phases = R, pf = arm == 2;
        void* args[] = {&a, &b, (void*)&chunk, &phases, (void*)&nb, &counter, &w, &pf, &sink};
        CK(hipLaunchCooperativeKernel((const void*)persistent, dim3(nb), dim3(256), args, 0, s));
      }
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    CK(hipMemcpy((arm ? h1 : h0).data(), a, (size_t)nb * chunk * 4, hipMemcpyDeviceToHost));
    bool same = true;
    if (arm) for (size_t i = 0; i < h0.size(); ++i) if (h0[i] != h1[i]) { same = false; break; }
    printf("arm %d: %.2f us per phase (best of %d, %d phases)%s\n", arm, best * 1000.f / R, reps, R,
           arm ? (same ? "  result == launches" : "  RESULT DIFFERS (coherence!)") : "");
  }
  return 0;
}
