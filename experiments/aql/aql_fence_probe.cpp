// Probe (round 6): what does a kernel boundary cost as a function of the AQL packet's acquire / release fence scopes?
// HIP submits every kernel of a stream with the barrier bit and agent- or system-scope fences; this probe writes the dispatch packets
// itself (ROCr user-mode queue, no HIP) and times chains of dependent phases -- each block reads what a block of ANOTHER XCD wrote in
// the previous phase -- for every (acquire, release) pair, comparing every result with the system/system chain bit for bit.
// build: g++ -O2 -I/opt/rocm/include aql_fence_probe.cpp -L/opt/rocm/lib -lhsa-runtime64 -o aql_fence_probe
// run:   ./aql_fence_probe phase_kernel.hsaco [chunk_floats]
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m = ""; hsa_status_string(s_, &m); printf("%s -> %s\n", #x, m); exit(1); } } while (0)

static hsa_agent_t gpu, cpu;
static hsa_amd_memory_pool_t dev_pool, karg_pool;
static bool have_gpu = false, have_cpu = false, have_dev = false, have_karg = false;

static hsa_status_t on_agent(hsa_agent_t a, void*) {
  hsa_device_type_t t;
  hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !have_gpu) { gpu = a; have_gpu = true; }
  if (t == HSA_DEVICE_TYPE_CPU && !have_cpu) { cpu = a; have_cpu = true; }
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t on_dev_pool(hsa_amd_memory_pool_t p, void*) {
  hsa_amd_segment_t seg;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  uint32_t fl;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
  if (seg == HSA_AMD_SEGMENT_GLOBAL && (fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !have_dev) { dev_pool = p; have_dev = true; }
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t on_cpu_pool(hsa_amd_memory_pool_t p, void*) {
  hsa_amd_segment_t seg;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  uint32_t fl;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
  if (seg == HSA_AMD_SEGMENT_GLOBAL && (fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !have_karg) { karg_pool = p; have_karg = true; }
  return HSA_STATUS_SUCCESS;
}

struct Args { const float* src; float* dst; int chunk; int phase; };

int main(int argc, char** argv) {
  if (argc < 2) { printf("usage: %s phase_kernel.hsaco [chunk_floats]\n", argv[0]); return 1; }
  const int chunk = argc > 2 ? atoi(argv[2]) : 64;
  CK(hsa_init());
  CK(hsa_iterate_agents(on_agent, nullptr));
  CK(hsa_amd_agent_iterate_memory_pools(gpu, on_dev_pool, nullptr));
  CK(hsa_amd_agent_iterate_memory_pools(cpu, on_cpu_pool, nullptr));
  if (!have_gpu || !have_dev || !have_karg) { printf("no gpu / pools\n"); return 1; }
  hsa_queue_t* q;
  CK(hsa_queue_create(gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));

  FILE* f = fopen(argv[1], "rb");
  if (!f) { printf("cannot open %s\n", argv[1]); return 1; }
  fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<char> co(sz);
  if (fread(co.data(), 1, sz, f) != (size_t)sz) return 1;
  fclose(f);
  hsa_code_object_reader_t rd;
  CK(hsa_code_object_reader_create_from_memory(co.data(), sz, &rd));
  hsa_executable_t exe;
  CK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
  CK(hsa_executable_load_agent_code_object(exe, gpu, rd, nullptr, nullptr));
  CK(hsa_executable_freeze(exe, nullptr));

  const size_t bytes = (size_t)256 * chunk * 4;
  float *a, *b;
  CK(hsa_amd_memory_pool_allocate(dev_pool, bytes, 0, (void**)&a));
  CK(hsa_amd_memory_pool_allocate(dev_pool, bytes, 0, (void**)&b));
  float* host;
  CK(hsa_amd_memory_pool_allocate(karg_pool, bytes, 0, (void**)&host));
  CK(hsa_amd_agents_allow_access(1, &gpu, nullptr, host));
  const int MAXP = 256;
  Args* kargs;
  CK(hsa_amd_memory_pool_allocate(karg_pool, sizeof(Args) * MAXP + 4096, 0, (void**)&kargs));
  CK(hsa_amd_agents_allow_access(1, &gpu, nullptr, kargs));
  Args* kargs_dev;      // the same table in device memory (what HIP_FORCE_DEV_KERNARG=1 does)
  CK(hsa_amd_memory_pool_allocate(dev_pool, sizeof(Args) * MAXP + 4096, 0, (void**)&kargs_dev));
  const bool dev_kernarg = !getenv("HOST_KERNARG");
  const int only = getenv("ONLY_WT") ? 1 : 0;
  hsa_signal_t done, cpy;
  CK(hsa_signal_create(1, 0, nullptr, &done));
  CK(hsa_signal_create(1, 0, nullptr, &cpy));

  auto copy = [&](void* dst, hsa_agent_t da, const void* src, hsa_agent_t sa, size_t nb = 0) {
    hsa_signal_store_relaxed(cpy, 1);
    CK(hsa_amd_memory_async_copy(dst, da, src, sa, nb ? nb : bytes, 0, nullptr, cpy));
    hsa_signal_wait_scacquire(cpy, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
  };

  const char* scope_name[3] = {"none", "agent", "system"};
  const int scope_val[3] = {HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_SYSTEM};
  printf("# kernel arguments in %s memory\n", dev_kernarg ? "device" : "host (fine-grained)");
  printf("# 256 blocks x 256 threads, chunk %d floats (%.2f KB per block and phase); per-phase time = (T(%d phases) - T(40 phases)) / %d, best of 20\n",
         chunk, chunk * 4 / 1024.0, 240, 200);
  for (int kv = only; kv < 2; ++kv) {
    const char* kname = kv ? "phase_wt.kd" : "phase_k.kd";
    hsa_executable_symbol_t sym;
    CK(hsa_executable_get_symbol_by_name(exe, kname, &gpu, &sym));
    uint64_t kobj;
    uint32_t kseg, gseg, pseg;
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kobj));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &kseg));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &gseg));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &pseg));
    if (kseg > sizeof(Args)) { printf("%s: kernarg segment %u > %zu (hidden arguments?)\n", kname, kseg, sizeof(Args)); }
    printf("## kernel %s (%s stores), kernarg %u B\n", kname, kv ? "write-through sc1" : "plain", kseg);
    std::vector<float> ref;
    for (int acq = 2; acq >= 0; --acq)
      for (int rel = 2; rel >= 0; --rel) {
        auto chain = [&](int phases, bool timed) -> double {
          for (int p = 0; p < phases; ++p) {
            kargs[p].src = (p & 1) ? b : a;
            kargs[p].dst = (p & 1) ? a : b;
            kargs[p].chunk = chunk;
            kargs[p].phase = p;
          }
          if (dev_kernarg) copy(kargs_dev, gpu, kargs, cpu, sizeof(Args) * MAXP);
          hsa_signal_store_relaxed(done, 1);
          const uint64_t first = hsa_queue_load_write_index_relaxed(q);
          for (int p = 0; p < phases; ++p) {
            const uint64_t idx = first + p;
            hsa_kernel_dispatch_packet_t* pk = (hsa_kernel_dispatch_packet_t*)q->base_address + (idx & (q->size - 1));
            memset((char*)pk + 4, 0, sizeof(*pk) - 4);
            pk->workgroup_size_x = 256; pk->workgroup_size_y = 1; pk->workgroup_size_z = 1;
            pk->grid_size_x = 256 * 256; pk->grid_size_y = 1; pk->grid_size_z = 1;
            pk->private_segment_size = pseg; pk->group_segment_size = gseg;
            pk->kernel_object = kobj;
            pk->kernarg_address = dev_kernarg ? &kargs_dev[p] : &kargs[p];
            const bool last = p == phases - 1, firstp = p == 0;
            pk->completion_signal.handle = last ? done.handle : 0;
            // first packet acquires at system scope (host wrote the kernargs / a previous chain's copy), last releases at system scope
            const int A = firstp ? HSA_FENCE_SCOPE_SYSTEM : scope_val[acq], R = last ? HSA_FENCE_SCOPE_SYSTEM : scope_val[rel];
            const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                    (A << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (R << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
            const uint16_t setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
            __atomic_store_n((uint32_t*)pk, (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
          }
          hsa_queue_store_write_index_release(q, first + phases);
          const auto t0 = std::chrono::steady_clock::now();
          hsa_signal_store_screlease(q->doorbell_signal, first + phases - 1);
          hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE);
          const auto t1 = std::chrono::steady_clock::now();
          (void)timed;
          return std::chrono::duration<double, std::micro>(t1 - t0).count();
        };
        // correctness: a and b start from a known state
        std::vector<float> init((size_t)256 * chunk);
        for (size_t i = 0; i < init.size(); ++i) init[i] = (float)(i % 97) * 0.01f;
        memcpy(host, init.data(), bytes);
        copy(a, gpu, host, cpu);
        copy(b, gpu, host, cpu);
        chain(40, false);
        copy(host, cpu, b, gpu);      // phase 39 (odd) wrote a; phase 38 wrote b -- fetch both
        std::vector<float> got(host, host + init.size());
        copy(host, cpu, a, gpu);
        got.insert(got.end(), host, host + init.size());
        bool same = true;
        if (ref.empty()) ref = got; else same = memcmp(ref.data(), got.data(), got.size() * 4) == 0;
        double t40 = 1e30, t240 = 1e30;
        for (int r = 0; r < 20; ++r) { t40 = std::min(t40, chain(40, true)); t240 = std::min(t240, chain(240, true)); }
        printf("acquire %-6s release %-6s: %.2f us per phase   (40 phases %.1f us, 240 phases %.1f us)   result %s\n", scope_name[acq],
               scope_name[rel], (t240 - t40) / 200.0, t40, t240, same ? "== system/system" : "DIFFERS (stale data)");
        fflush(stdout);
      }
  }
  return 0;
}
