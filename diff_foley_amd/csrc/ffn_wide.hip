// WIDE-TILE GEGLU projection (round 6): the first Linear of FeedForward (attention_openai.py:37-64: Linear(C -> 8C), x * gelu(gate))
// with the pre-norm LayerNorm (norm3, attention_openai.py:215) folded in, like ffn.hip -- on BM x 320 output tiles.
//
//   out[m][o] = xh * gelu(gh),   xh = rstd[m] * (A[m] . Wx[o] - mean[m] * cs[x_o]) + b[x_o]   (gh likewise with the gate row)
//
// Why another kernel.  The persistent 128 x 128 form (ffn.hip) moves (128 + 128) x 128 B through the L2 -> LDS fill path per 512
// cycles of MFMA work: 64 B/clk per CU asked of a path that delivers ~32 (profiles/r5_pgeglu_stamps.txt: ~2000 cycles per K step
// with two resident blocks, i.e. the matrix pipe half idle inside the loop), and 1280 tiles on 512 resident blocks are 2.5 tiles
// per block -- half the blocks run 3.  The model's channel counts are 5 * 2^k, so N = 8C is a multiple of 320 at every level, and
//      (M, N) = (8192, 2560) / (2048, 5120) / (512, 10240)  are EXACTLY 256 tiles of 256 / 128 / 64 rows x 320 columns:
// one tile per CU, one round, and (BM + 320) x 128 B per K step for BM x 320 x 64 MACs -- 29 / 45 / 77 B/clk at the full MFMA rate
// against 64 before (the 64-row form is weight-streaming bound whatever the tile).
//
// Block = 8 wavefronts, 4 (M) x 2 (N), two per SIMD: while one wavefront's MFMAs run, its SIMD partner issues fragment reads and
// operand requests.  A wavefront owns 16 RT rows x 160 columns as RT x 10 fragments of v_mfma_f32_16x16x32 (RT = 4 / 2 / 1).
// The product is formed TRANSPOSED (weights as the MFMA's row operand): a lane then holds 4 CONSECUTIVE output columns of one
// row per fragment, x and gate of an output in the same lane and register of fragments j and j + 5 (the 320-column weight
// packing, GemmParams::W_w320: every 160-row half of a tile is [80 x rows | their 80 gate rows]) -- the LayerNorm fold needs one
// (rstd, -rstd mean) pair per lane and row fragment, the GELU result leaves as 8-byte LDS stores, and the tile goes out as full
// 320-byte row segments in 16-byte write-through stores.
// Operands: HBM / L2 -> LDS by buffer_load ... lds (XOR-swizzled through the source address like gemm_impl.h), NST stages, one raw
// s_barrier per K step with lgkmcnt(0) + vmcnt(0) in front of it: the request of step k + 1 is issued at the top of step k and has
// the whole step (>= 1280 cycles per wavefront pair) to land.
#include <type_traits>

#include "gemm_impl.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int RT, int NST, bool DBG>
__global__ __launch_bounds__(512, 2) void geglu_wide_kernel(GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  const DfTouch ka = gemm_kernarg_touch();
  df_fp16_hw_clamp();                      // the epilogue packs with pack_bf2_hw
  constexpr int BM = 64 * RT, BN = 320, NT = 512, RPP = NT / 8;      // RPP = 64 LDS rows per request pass of the block
  constexpr int AP = BM / RPP, BP = BN / RPP;                          // request instructions per wavefront and K step: RT + 5
  constexpr int STAGE = (BM + BN) * BK * 2;                            // bytes of one stage: [A tile | W tile], 128-byte rows
  constexpr int OSTR = 336;                                            // bytes per row of the epilogue's output patch (160 outputs + pad)
  static_assert(BM % RPP == 0 && BN % RPP == 0 && NST >= 2 && BM * OSTR <= NST * STAGE, "tile shape");
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  float2* const sRow = reinterpret_cast<float2*>(smem + NST * STAGE);            // [BM] (rstd, -rstd * mean)
  float* const sCs = reinterpret_cast<float*>(smem + NST * STAGE + BM * 8);      // [320] column sums, then [320] folded bias
  float* const sBb = sCs + BN;

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  int n_stamp = 0;
  auto stamp = [&]() {
    if (DBG && (p.dbg & 64) && tid == 0 && n_stamp < 32)
      reinterpret_cast<unsigned long long*>(p.partial)[(long)blockIdx.x * 32 + n_stamp++] = __builtin_amdgcn_s_memtime();
  };
  stamp();

  // ---- tile id: each XCD owns a contiguous range of logical tiles (block b runs on XCD b % 8), walked by p.gm like the generic kernel
  const int nbm = (p.M + BM - 1) / BM, nbn = p.N / BN, nblk = nbm * nbn;
  int lid;
  {
    const int bid = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int mt, nt;
  tile_of(lid, nbm, nbn, p.gm, mt, nt);
  const int m0 = mt * BM, n0 = nt * BN;
  const int nk = p.K / BK;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W_w320, 0, (int)p.w_bytes, 0x00020000);
  const int r0 = tid >> 3;
  const int c8 = ((tid & 7) ^ ((r0 >> 1) & 7)) * 8;          // source chunk that lands in LDS slot (tid & 7) of rows r0 + 64 i
  unsigned a_off[AP], b_off[BP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int m = m0 + r0 + RPP * i;
    a_off[i] = m < p.M ? (unsigned)(((long)m * p.lda + c8) * 2) : OOB;
  }
#pragma unroll
  for (int i = 0; i < BP; ++i) b_off[i] = (unsigned)(((long)(n0 + r0 + RPP * i) * p.K + c8) * 2);
  char* const dma0 = smem + wid * (8 * BK * 2);
  // one stage: every wavefront writes 8 rows x 128 B (1 KiB, lane-linear) per instruction.  Stages past the end of K are requested
  // out of bounds (the hardware writes zeros into a slot nobody reads), so the request count per step is constant and the waits
  // stay counted.
  // part I of the AP + BP request instructions of stage kt into ring slot `slot` (I < AP: activation rows, else weight rows)
  auto dma_part = [&](auto part, int kt, int slot) {
    constexpr int I = decltype(part)::value;
    if (DBG && (p.dbg & 8)) return;
    const bool live = kt < nk;
    const unsigned kb = (unsigned)kt * (BK * 2);
    char* const dst = dma0 + slot * STAGE;
    if constexpr (I < AP)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(dst + I * RPP * (BK * 2)), 16, (live && a_off[I] != OOB) ? a_off[I] + kb : OOB, 0, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(dst + (BM + (I - AP) * RPP) * (BK * 2)), 16, live ? b_off[I - AP] + kb : OOB, 0, 0, 0);
  };
#define WG_DMA(I, KT, SLOT) \
  if constexpr ((I) < AP + BP) dma_part(std::integral_constant<int, (I) < AP + BP ? (I) : 0>{}, KT, SLOT)
  auto dma_step = [&](int kt, int slot) {
    WG_DMA(0, kt, slot); WG_DMA(1, kt, slot); WG_DMA(2, kt, slot); WG_DMA(3, kt, slot); WG_DMA(4, kt, slot);
    WG_DMA(5, kt, slot); WG_DMA(6, kt, slot); WG_DMA(7, kt, slot); WG_DMA(8, kt, slot);
  };

  // ---- epilogue operands, requested FIRST (older than every operand request: the compiler's wait for them is a counted one, not
  // a drain of the stages): LayerNorm row statistics of tile row `tid` (threads < BM: ln_slots = C / 64 = 5, 10 or 20 partial
  // (sum, sumsq) pairs, folded below) and this tile's column sums / folded bias (threads < 320)
  constexpr int LNS = 20;
  float2 lnv[LNS];
  const bool has_row = tid < BM;
  if (has_row && (!DBG || !(p.dbg & 32))) {
    const float2* sp = p.ln_stats + (long)min(m0 + tid, p.M - 1) * p.ln_slots;
#pragma unroll
    for (int i = 0; i < LNS; ++i) lnv[i] = sp[min(i, p.ln_slots - 1)];      // (clamped: no per-load branch, the surplus loads hit the same line)
  }
  float ccs = 0.f, cbb = 0.f;
  if (tid < BN) {
    ccs = p.cs_w320[n0 + tid];
    cbb = p.bias_w320[n0 + tid];
  }

  // ---- prologue: the first NST - 1 stages are requested before anything else is computed
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) dma_step(s, s);
  __builtin_amdgcn_sched_barrier(0);
  stamp();

  // (the loaded pairs pass through an empty asm HERE: without it the compiler hoists the first add of the fold above the operand
  // requests and the block waits for a statistics load -- and, loads retiring in order, for the entry touch -- before it has
  // requested anything)
  if (has_row && (!DBG || !(p.dbg & 32))) {
#pragma unroll
    for (int i = 0; i < LNS; ++i) asm volatile("" : "+v"(lnv[i].x), "+v"(lnv[i].y));
  }
  if (has_row) {
    float s1 = 0.f, s2 = 0.f;
    if (!DBG || !(p.dbg & 32)) {
#pragma unroll
      for (int i = 0; i < LNS; ++i)
        if (i < p.ln_slots) {
          s1 += lnv[i].x;
          s2 += lnv[i].y;
        }
    }
    const float inv = 1.0f / (float)p.ln_C;
    const float mean = s1 * inv;
    const float rstd = rsqrtf(fmaxf(s2 * inv - mean * mean, 0.f) + p.ln_eps);
    sRow[tid] = make_float2(rstd, -rstd * mean);
  }
  if (tid < BN) {
    sCs[tid] = ccs;
    sBb[tid] = cbb;
  }

  // ---- LDS fragment byte offsets inside a stage.  16 x 16 x 32 fragments: lane -> row l15 of the fragment, 16-byte chunk 4 s + l4 of
  // its 128-byte row; the swizzle term (row >> 1) & 7 is (l15 >> 1) for every fragment (fragment rows start at multiples of 16), so
  // fragment j / rt is a compile-time offset of 2 KiB per fragment from two per-lane bases per operand.
  int fA0[2], fW0[2];
  {
    const int rowa = wm * (16 * RT) + l15, roww = wn * 160 + l15, sw = (l15 >> 1) & 7;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      fA0[s] = rowa * (BK * 2) + (((4 * s + l4) ^ sw) << 4);
      fW0[s] = BM * (BK * 2) + roww * (BK * 2) + (((4 * s + l4) ^ sw) << 4);
    }
  }

  f32x4_t acc[10][RT];
#pragma unroll
  for (int j = 0; j < 10; ++j)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[j][rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // stage 0 landed (my requests: the counted wait; everybody's: the barrier); the LDS tables are ordered by the same barrier
  constexpr int VM = (NST - 2) * (AP + BP);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  wait_vmcnt<VM>();
  __builtin_amdgcn_s_barrier();
  stamp();

  // One K step = four quarters (k half s, column half h): 5 RT MFMAs each out of (RT + 5 | 5) fragments.  The fragments of quarter
  // q + 1 are read while quarter q's MFMAs run (two fragment sets: afA / afB for the two k halves, wfA / wfB for the column halves --
  // 72 registers for RT = 4), and the step's RT + 5 operand requests are dealt out between the quarters instead of standing in
  // front of the first fragment read (an LDS-DMA request costs its wavefront ~100 issue cycles: nine in a row were ~900 cycles
  // per step in which neither wavefront of a SIMD had an MFMA to issue; tools/pgeglu_stamps.py: 4.3 k cycles per step against
  // 2.56 k of MFMA before this schedule).
#define WG_LOAD_A(DST, S)                                                                                          \
  if (!DBG || !(p.dbg & 16)) {                                                                                     \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) DST[rt] = *reinterpret_cast<const bf16x8*>(rbase + fA0[S] + rt * 16 * (BK * 2)); \
  }
#define WG_LOAD_W(DST, S, H)                                                                                       \
  if (!DBG || !(p.dbg & 16)) {                                                                                     \
    _Pragma("unroll") for (int j = 0; j < 5; ++j) DST[j] = *reinterpret_cast<const bf16x8*>(rbase + fW0[S] + (5 * (H) + j) * 16 * (BK * 2)); \
  }
#define WG_MMA(WF, AF, H)                                                                                          \
  if (!DBG || !(p.dbg & 1)) {                                                                                      \
    _Pragma("unroll") for (int j = 0; j < 5; ++j)                                                                  \
      _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) acc[5 * (H) + j][rt] = DF_MFMA_16x16x32(WF[j], AF[rt], acc[5 * (H) + j][rt]); \
  } else {                                                                                                         \
    _Pragma("unroll") for (int j = 0; j < 5; ++j) asm volatile("" ::"v"(WF[j]));                                   \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) asm volatile("" ::"v"(AF[rt]));                              \
  }
  int slot = 0;
#pragma unroll 1
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt + NST - 1 goes into the slot step kt - 1 read: every wavefront finished those reads before the barrier it just passed
    const int nslot = (slot + NST - 1 >= NST) ? slot - 1 : slot + NST - 1, nkt = kt + NST - 1;
    const char* const rbase = smem + slot * STAGE;
    bf16x8 afA[RT], afB[RT], wfA[5], wfB[5];
    WG_LOAD_A(afA, 0);
    WG_LOAD_W(wfA, 0, 0);
    WG_LOAD_W(wfB, 0, 1);
    WG_DMA(0, nkt, nslot); WG_DMA(1, nkt, nslot); WG_DMA(2, nkt, nslot);
    __builtin_amdgcn_sched_barrier(0);
    WG_MMA(wfA, afA, 0);
    __builtin_amdgcn_sched_barrier(0);
    WG_LOAD_A(afB, 1);
    WG_LOAD_W(wfA, 1, 0);
    WG_DMA(3, nkt, nslot); WG_DMA(4, nkt, nslot);
    __builtin_amdgcn_sched_barrier(0);
    WG_MMA(wfB, afA, 1);
    __builtin_amdgcn_sched_barrier(0);
    WG_LOAD_W(wfB, 1, 1);
    WG_DMA(5, nkt, nslot); WG_DMA(6, nkt, nslot);
    __builtin_amdgcn_sched_barrier(0);
    WG_MMA(wfA, afB, 0);
    __builtin_amdgcn_sched_barrier(0);
    WG_DMA(7, nkt, nslot); WG_DMA(8, nkt, nslot);
    __builtin_amdgcn_sched_barrier(0);
    WG_MMA(wfB, afB, 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // my reads of this slot are done: it may be refilled after the barrier
    wait_vmcnt<VM>();                                          // my pieces of stage kt + 1 landed (<= NST - 2 younger stages in flight)
    __builtin_amdgcn_s_barrier();
    slot = (slot + 1 == NST) ? 0 : slot + 1;
    stamp();
  }
#undef WG_LOAD_A
#undef WG_LOAD_W
#undef WG_MMA
  if constexpr (NST > 2) {      // dead-slot requests of the last steps land before the output patch overlays the stages
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue out of the accumulators.  Transposed product: lane (l15, l4) of fragment (j, rt) holds output row
  // wm * 16 RT + rt * 16 + l15, tile columns wn * 160 + 16 j + 4 l4 + (0..3); x in fragments 0..4, its gate in 5..9.
  // (every wavefront passed the last K-step barrier: no stage is read any more -- the output patch may overlay the stages)
  char* const patch = smem;
  float2 mr[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) mr[rt] = sRow[wm * (16 * RT) + rt * 16 + l15];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int cx = wn * 160 + 16 * j + 4 * l4, cg = cx + 80;
    const float4 csx = *reinterpret_cast<const float4*>(&sCs[cx]), csg = *reinterpret_cast<const float4*>(&sCs[cg]);
    const float4 bbx = *reinterpret_cast<const float4*>(&sBb[cx]), bbg = *reinterpret_cast<const float4*>(&sBb[cg]);
    const float cxs[4] = {csx.x, csx.y, csx.z, csx.w}, cgs[4] = {csg.x, csg.y, csg.z, csg.w};
    const float bxs[4] = {bbx.x, bbx.y, bbx.z, bbx.w}, bgs[4] = {bbg.x, bbg.y, bbg.z, bbg.w};
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = wm * (16 * RT) + rt * 16 + l15;
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float xv = acc[j][rt][r] * mr[rt].x + (mr[rt].y * cxs[r] + bxs[r]);
        const float gv = acc[j + 5][rt][r] * mr[rt].x + (mr[rt].y * cgs[r] + bgs[r]);
        o[r] = (DBG && (p.dbg & 4)) ? xv + gv : xv * gelu_erf(gv);
      }
      *reinterpret_cast<uint2*>(patch + row * OSTR + (wn * 80 + 16 * j + 4 * l4) * 2) = make_uint2(pack_bf2_hw(o[0], o[1]), pack_bf2_hw(o[2], o[3]));
    }
  }
  stamp();
  // ---- stores.  A wavefront reads back ITS OWN 16 RT x 80 patch (no block barrier: its lgkmcnt(0) orders its own LDS writes in
  // front of its reads) and sends it out as 160-byte row segments in 16-byte write-through stores -- the wavefronts drift apart and
  // one's stores run beside another's GELU arithmetic.
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  {
    bf16_t* const Cb = reinterpret_cast<bf16_t*>(p.C);
    const int ocol0 = nt * (BN / 2) + wn * 80, prow0 = wm * (16 * RT);
    constexpr int CH = 16 * RT * 10;                 // 16-byte chunks of the wavefront's patch
#pragma unroll
    for (int it = 0; it < (CH + 63) / 64; ++it) {
      const int e = lane + it * 64, row = e / 10, ch = e - row * 10;
      if (CH % 64 != 0 && e >= CH) break;
      const uint4 v = *reinterpret_cast<const uint4*>(patch + (prow0 + row) * OSTR + wn * 160 + ch * 16);
      if (m0 + prow0 + row < p.M && (!DBG || !(p.dbg & 2)))
        st_wt(reinterpret_cast<uint4*>(Cb + (long)(m0 + prow0 + row) * p.ldc + ocol0 + ch * 8), v);
    }
  }
  stamp();
  gemm_kernarg_touch_end(ka);
#endif
}

template <int RT, int NST, bool DBG>
hipError_t launch_wgeglu(const GemmParams& p, hipStream_t stream) {
  constexpr int BM = 64 * RT;
  constexpr size_t lds = (size_t)NST * (BM + 320) * BK * 2 + (size_t)BM * 8 + (size_t)2 * 320 * 4;
  static_assert(lds <= 160 * 1024, "LDS");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&geglu_wide_kernel<RT, NST, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int nblk = ((p.M + BM - 1) / BM) * (p.N / 320);
  hipLaunchKernelGGL((geglu_wide_kernel<RT, NST, DBG>), dim3(nblk), dim3(512), lds, stream, p);
  return hipGetLastError();
}

// rows of the (32 x | 32 gate) packing -> the 320-column packing of the wide tiles; one thread per 16-byte chunk of a weight row,
// thread (row, 0) also moves that row's column sum and folded bias
__global__ __launch_bounds__(256) void pack_w320_kernel(const uint16_t* __restrict__ w, const float* __restrict__ cs, const float* __restrict__ bb,
                                                        uint16_t* __restrict__ wo, float* __restrict__ cso, float* __restrict__ bbo, int N, int K) {
  const int kc = K / 8;
  const long total = (long)N * kc;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int prow = (int)(e / kc), c = (int)(e - (long)prow * kc);
    const int grp = prow >> 6, within = prow & 63, is_gate = within >> 5, og = grp * 32 + (within & 31);      // output column of this row
    const int t = og / 160, o = og - t * 160, h = o / 80;
    const int nrow = t * 320 + h * 160 + is_gate * 80 + (o - h * 80);
    reinterpret_cast<uint4*>(wo)[(long)nrow * kc + c] = reinterpret_cast<const uint4*>(w)[e];
    if (c == 0) {
      cso[nrow] = cs[prow];
      bbo[nrow] = bb[prow];
    }
  }
}

}  // namespace

hipError_t launch_pack_w320(const uint16_t* w, const float* cs, const float* bb, uint16_t* wo, float* cso, float* bbo, int N, int K, hipStream_t s) {
  if (N % 320 != 0 || K % 8 != 0 || N % 64 != 0) return hipErrorInvalidValue;
  const long total = (long)N * (K / 8);
  const int blocks = (int)std::min<long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(pack_w320_kernel, dim3(blocks), dim3(256), 0, s, w, cs, bb, wo, cso, bbo, N, K);
  return hipGetLastError();
}

bool wgeglu_valid(const GemmParams& p, int tile, int batch, int splitk) {
  return p.geglu && p.ln_stats && p.W_w320 && p.cs_w320 && p.bias_w320 && splitk <= 1 && batch <= 1 && p.taps == 1 && p.out_bf16 && !p.res &&
         !p.rowbias && !p.aux && !p.stats && !p.vt && p.w_rows == 0 && p.Cin2 == 0 && p.dup_rows == 0 && p.sm_w == 0 && !p.relu && !p.silu &&
         !p.store_nchw && p.alpha == 1.f && (p.N % 320) == 0 && (p.K % 64) == 0 && p.K >= 128 && p.ln_slots <= 20 && (p.ln_slots % 5) == 0 &&
         p.C != nullptr && (p.ldc % 8) == 0 && (p.lda % 8) == 0;
}

hipError_t launch_gemm_wgeglu(int tile_cfg, const GemmParams& p, hipStream_t stream) {
  switch (tile_cfg) {
    case TILE_WGEGLU_256: return p.dbg ? launch_wgeglu<4, 2, true>(p, stream) : launch_wgeglu<4, 2, false>(p, stream);
    case TILE_WGEGLU_128: return p.dbg ? launch_wgeglu<2, 2, true>(p, stream) : launch_wgeglu<2, 2, false>(p, stream);
    case TILE_WGEGLU_64: return p.dbg ? launch_wgeglu<1, 3, true>(p, stream) : launch_wgeglu<1, 3, false>(p, stream);
    default: return hipErrorInvalidValue;
  }
}
