// Persistent GEGLU projection (the first Linear of FeedForward, attention_openai.py:37-64: Linear(C -> 8C), x * gelu(gate)), with
// the pre-norm LayerNorm (norm3, attention_openai.py:215) folded in:
//
//   out[m][o] = xh * gelu(gh),   xh = rstd[m] * (A[m] . Wx[o] - mean[m] * cs[x_o]) + b[x_o]   (gh likewise with the gate row)
//
// Why its own kernel.  st.ff1 is the largest GEMM family of a denoise step (16 ops, 13.4 GFLOP each, 0.55 ms of 4.1) and the
// generic kernel ran it at 350-395 TFLOP/s whatever the tile (profiles/r3_op_decomposition.csv: 12 us "floor" for its 2560 blocks +
// 14 us loop + 12 us epilogue at M = 8192).  K is short (5-20 K steps), so a block's life was: compute offsets, fill the ring
// (one exposed L2 round trip), 5-20 steps, drain, park the tile in LDS, barrier, GEGLU, store -- the operand ring carried nothing
// through a third of it.  Here:
//   * 2 resident blocks per CU walk a tile queue (grid = 2 x 256): kernel arguments, descriptors and the instruction cache are
//     paid once per CU, not once per tile;
//   * the (tile, k) sequence of a block is ONE continuous request stream -- the first stages of tile i+1 are in flight while tile
//     i finishes and runs its epilogue, so the ring never drains (LDS slots are addressed dynamically: K / 64 is not a multiple of
//     the ring depth);
//   * the epilogue runs out of the accumulator registers: x and gate of one output sit in the same lane of adjacent MFMA tiles
//     (weights are packed in (32 x | 32 gate) row groups), the LayerNorm row statistics come from a 1 KB LDS table -- no parking
//     of the fp32 tile, no block barrier; the other resident block's MFMAs run beside this block's GELU arithmetic (separate
//     pipes of the SIMD), and the row statistics / column vectors of the NEXT tile are requested before the GELU and folded
//     after it.
// MFMA: v_mfma_f32_32x32x16_{bf16,f16}; operands HBM/L2 -> LDS by buffer_load ... lds (no VGPR round trip), XOR-swizzled through
// the source address like gemm_impl.h; counted vmcnt, one raw s_barrier per K step, lgkmcnt(0) in front of it (ring hand-back).
#include "gemm_impl.h"

namespace {

// BM x 128 tile, WGM x WGN wavefronts (4 or 8); every wavefront owns 32 rows x (128 / WGN) columns.  Two resident blocks per CU.
template <int BM, int WGM, int WGN, int NST, int LNS, bool DBG>
__device__ __forceinline__ void geglu_persistent_body(const GemmParams& p);
template <int BM, int WGM, int WGN, int NST, int LNS, bool DBG>
__global__ __launch_bounds__(64 * WGM * WGN, WGM * WGN / 2) void geglu_persistent_kernel(GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  const DfTouch ka = gemm_kernarg_touch();
  df_fp16_hw_clamp();           // the epilogue packs with pack_bf2_hw
  geglu_persistent_body<BM, WGM, WGN, NST, LNS, DBG>(p);
  gemm_kernarg_touch_end(ka);
#endif
}
template <int BM, int WGM, int WGN, int NST, int LNS, bool DBG>
__device__ __forceinline__ void geglu_persistent_body(const GemmParams& p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BN = 128, NT = 64 * WGM * WGN, RPP = NT / 8;
  constexpr int WTN = BN / WGN, TN = WTN / 32, NG = TN / 2;     // NG (x | gate) group pairs per wavefront
  constexpr int AP = BM / RPP, BP = BN / RPP, LPT = AP + BP;
  constexpr int STAGE = (BM + BN) * BK * 2;                      // bytes of one ring stage: [A tile | W tile]
  static_assert(BM == 32 * WGM && TN % 2 == 0 && (WGM * WGN == 4 || WGM * WGN == 8) && BM % RPP == 0 && BN % RPP == 0, "tile shape");
  static_assert(32 * (32 * NG * 2) <= LPT * 1024, "the epilogue's transposition patch fits this wavefront's own request pieces of a stage");
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  float2* sRow = reinterpret_cast<float2*>(smem + NST * STAGE);     // [2][BM] (rstd, -rstd * mean) of the current / next tile's rows

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WGN, wn = wid % WGN;
  const int l31 = lane & 31, lh = lane >> 5;
  // DBG + p.dbg bit 6: thread 0 of every block stamps the shader clock at phase boundaries into p.partial[block][32] (uint64)
  int n_stamp = 0;
  auto stamp = [&]() {
    if (DBG && (p.dbg & 64) && tid == 0 && n_stamp < 32)
      reinterpret_cast<unsigned long long*>(p.partial)[(long)blockIdx.x * 32 + n_stamp++] = __builtin_amdgcn_s_memtime();
  };
  stamp();

  // ---- this block's tile list: each XCD owns a contiguous range of logical tiles (its L2 shares their operand panels); the
  // blocks of an XCD take the range's tiles round-robin
  const int nbm = (p.M + BM - 1) / BM, nbn = p.N / BN, nblk = nbm * nbn;
  const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
  const int q8 = nblk >> 3, r8 = nblk & 7;
  const int ts = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int cnt = q8 + (xcd < r8 ? 1 : 0);
  const int nbx = ((int)gridDim.x - xcd + 7) >> 3;               // blocks resident on this XCD
  const int mine = idx < cnt ? (cnt - idx + nbx - 1) / nbx : 0;  // tiles of this block
  const int nk = p.K / BK;
  if (mine == 0) return;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)p.w_bytes, 0x00020000);
  // output through a buffer descriptor: a row that does not exist gets an out-of-bounds offset and the hardware drops the store
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)((long)p.M * p.ldc * 2), 0x00020000);
  const int r0 = tid >> 3;
  const int c8 = ((tid & 7) ^ ((r0 >> 1) & 7)) * 8;             // source chunk that lands in LDS slot (tid & 7) of rows r0 + 32 i

  // ---- request stream state: (tile ordinal, k tile) of the NEXT stage to request, its operand offsets, its ring slot
  int d_i = 0, d_k = 0, d_slot = 0;
  unsigned a_off[AP], b_off[BP];
  auto tile_mn = [&](int i, int& m0, int& n0) {
    int mt, nt;
    tile_of(ts + idx + i * nbx, nbm, nbn, p.gm, mt, nt);
    m0 = mt * BM;
    n0 = nt * BN;
  };
  auto set_dma_tile = [&](int i) {
    int m0, n0;
    tile_mn(min(i, mine - 1), m0, n0);
    const bool live = i < mine;
#pragma unroll
    for (int j = 0; j < AP; ++j) {
      const int m = m0 + r0 + RPP * j;
      a_off[j] = (live && m < p.M) ? (unsigned)(((long)m * p.lda + c8) * 2) : OOB;
    }
#pragma unroll
    for (int j = 0; j < BP; ++j) b_off[j] = live ? (unsigned)(((long)(n0 + r0 + RPP * j) * p.K + c8) * 2) : OOB;
  };
  char* const dma0 = smem + wid * (8 * BK * 2);
  // one stage: every wavefront writes 8 rows x 128 B (1 KiB, lane-linear) per instruction; stages past the end of the list are
  // requested out of bounds (zeros land in a dead slot), so the request count per step is constant and the waits stay counted
  auto dma_step = [&]() {
    const unsigned kb = (unsigned)d_k * (BK * 2);
    char* const dst = dma0 + d_slot * STAGE;
    if (!DBG || !(p.dbg & 8)) {
#pragma unroll
    for (int j = 0; j < AP; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(dst + j * RPP * (BK * 2)), 16, a_off[j] == OOB ? OOB : a_off[j] + kb, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < BP; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(dst + (BM + j * RPP) * (BK * 2)), 16, b_off[j] == OOB ? OOB : b_off[j] + kb, 0, 0, 0);
    }
    d_slot = (d_slot + 1 == NST) ? 0 : d_slot + 1;
    if (++d_k == nk) {
      d_k = 0;
      ++d_i;
      set_dma_tile(d_i);
    }
  };

  // ---- prologue: NST-1 stages in flight before anything else is computed
  set_dma_tile(0);
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) dma_step();
  __builtin_amdgcn_sched_barrier(0);
  stamp();

  // LDS fragment byte offsets inside a stage for the 4 k-steps of a tile
  int fA[4], fB[TN][4];
  {
    const int row = wm * 32 + l31;
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) fA[s2] = (row * BK + (((2 * s2 + lh) ^ ((row >> 1) & 7)) << 3)) * 2;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * WTN + j * 32 + l31;
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) fB[j][s2] = BM * BK * 2 + (row * BK + (((2 * s2 + lh) ^ ((row >> 1) & 7)) << 3)) * 2;
  }

  // per-tile epilogue operands: LayerNorm row statistics of tile row `tid` (threads < BM) and this lane's column vectors.
  // request(i) issues the loads for tile ordinal i, fold(i) turns them into sRow[i & 1][] / keeps the column vectors.
  // Row statistics: TPR = NT / BM neighbouring lanes share a tile row, lane `sub` takes slots sub, sub + TPR, ... (<= LPS per lane:
  // 3-5 registers pairs ride through the K loop instead of 10-20) and the lanes' sums meet in a quad exchange.
  constexpr int TPR = NT / BM, LPS = (LNS + TPR - 1) / TPR;
  static_assert(TPR == 2 || TPR == 4, "lanes per tile row");
  const int srow_ = tid / TPR, ssub = tid % TPR;
  float2 lnv[LPS];
  float ccs[TN], cbb[TN];
  auto request = [&](int i) {
    if (DBG && (p.dbg & 32)) return;
    int m0, n0;
    tile_mn(min(i, mine - 1), m0, n0);
    {
      const float2* sp = p.ln_stats + (long)min(m0 + srow_, p.M - 1) * p.ln_slots;
#pragma unroll
      for (int s = 0; s < LPS; ++s) {
        const int slot = ssub + s * TPR;
        lnv[s] = slot < p.ln_slots ? sp[slot] : make_float2(0.f, 0.f);
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * WTN + j * 32 + l31;
      ccs[j] = p.ln_cs[col];
      cbb[j] = p.bias[col];
    }
  };
  float ecs[TN], ebb[TN];        // column vectors of the tile being computed (ccs / cbb hold the next tile's)
  auto fold = [&](int i) {
    if (DBG && (p.dbg & 32)) return;
    {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int s = 0; s < LPS; ++s) {
        s1 += lnv[s].x;
        s2 += lnv[s].y;
      }
      s1 += dpp_xor1(s1);            // DPP operands: no ds_bpermute round trips in the fold
      s2 += dpp_xor1(s2);
      if (TPR == 4) {
        s1 += dpp_xor2(s1);
        s2 += dpp_xor2(s2);
      }
      const float inv = 1.0f / (float)p.ln_C;
      const float mean = s1 * inv;
      const float rstd = rsqrtf(fmaxf(s2 * inv - mean * mean, 0.f) + p.ln_eps);
      if (ssub == 0) sRow[(i & 1) * BM + srow_] = make_float2(rstd, -rstd * mean);     // the epilogue's FMA operands
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      ecs[j] = ccs[j];
      ebb[j] = cbb[j];
    }
  };
  request(0);
  fold(0);                         // the only exposed round trip of these operands (the prologue stages are needed next anyway)
  if (mine > 1) request(1);        // from here on: requested a whole tile before they are folded

  // VM = vector-memory operations of this wave that may still be in flight = the requests of the NST-2 youngest stages.  Loads
  // retire in issue order, so "at most VM outstanding" implies every older stage has landed; the epilogue's stores sit in the same
  // counter and may retire out of order with the loads, which only makes the wait stricter (it drains them), never weaker
#define PG_RING_SYNC(VM)                                                                         \
  {                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* my reads of the slot about to be refilled are done */ \
    wait_vmcnt<(VM)>();                                                                         \
    __builtin_amdgcn_s_barrier();                                                               \
  }
#define PG_FRAG(DA, DB, S)                                                                        \
  if (!DBG || !(p.dbg & 16)) {                                                                  \
    DA = *reinterpret_cast<const bf16x8*>(rbase + fA[S]);                                       \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) DB[j] = *reinterpret_cast<const bf16x8*>(rbase + fB[j][S]); \
  }
  // DBG instantiation only (tools/pgeglu_probe.py): p.dbg bit 0 no MFMAs, bit 1 no stores, bit 2 no GELU, bit 3 no operand requests,
  // bit 4 no fragment reads, bit 5 no row-statistics / column-vector loads
#define PG_MMA(SA, SB)                                                                            \
  {                                                                                             \
    if (!DBG || !(p.dbg & 1)) {                                                                 \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[j] = DF_MFMA_32x32x16(SA, SB[j], acc[j]); \
    } else {                                                                                    \
      asm volatile("" ::"v"(SA));                                                               \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(SB[j]));            \
    }                                                                                           \
  }

  int r_slot = 0;
  constexpr int VM_STEADY = (NST - 2) * LPT;
  stamp();
  PG_RING_SYNC(VM_STEADY);         // stage 0 landed and visible (also orders sRow[0])
  stamp();
  for (int ti = 0; ti < mine; ++ti) {
    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int kk = 0; kk < nk; ++kk) {
      const char* const rbase = smem + r_slot * STAGE;
      bf16x8 a0, a1, b0[TN], b1[TN];
      PG_FRAG(a0, b0, 0);
      dma_step();                  // stage (current + NST - 1) into the slot the previous step read
      PG_FRAG(a1, b1, 1);
      PG_MMA(a0, b0);
      PG_FRAG(a0, b0, 2);
      PG_MMA(a1, b1);
      PG_FRAG(a1, b1, 3);
      PG_MMA(a0, b0);
      PG_RING_SYNC(VM_STEADY);
      PG_MMA(a1, b1);
      r_slot = (r_slot + 1 == NST) ? 0 : r_slot + 1;
      stamp();
    }
    // ---- epilogue of tile ti out of the accumulators.  The row statistics / column vectors of tile ti+1 were requested one tile
    // ago (they landed during this tile's K loop): fold them first, request tile ti+2's, then the GEGLU arithmetic and the stores
    // -- a wait for those loads is never a wait for this epilogue's stores
    int m0, n0;
    tile_mn(ti, m0, n0);
    float xcs[TN], xbb[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      xcs[j] = ecs[j];
      xbb[j] = ebb[j];
    }
    if (ti + 1 < mine) fold(ti + 1);
    if (ti + 2 < mine) request(ti + 2);
    stamp();
    // LayerNorm fold:  v = acc * rstd + (b - rstd * mean * cs).  Plain fp32 FMAs: the packed forms (v_pk_fma_f32) measured slower
    // here (5.1 k vs ~3.5 k cycles for this block of arithmetic) -- fp32 VALU work is bound by lanes per clock, not by issue slots
    // (Two halves of 8 accumulator rows: only 8 rows' (rstd, -rstd * mean) pairs are live at a time -- the 8-wavefront form runs at
    // 4 wavefronts per SIMD, 128 registers.)
    const float2* sR = sRow + (ti & 1) * BM + wm * 32;
    uint32_t o32[NG][8];           // outputs of accumulator rows (2q, 2q+1) as operand-type pairs
#define PG_HALF(H)                                                                                \
    {                                                                                           \
      float rs[8], rm[8];                                                                       \
      _Pragma("unroll") for (int r8 = 0; r8 < 8; ++r8) {                                        \
        const int r = 8 * (H) + r8;                                                             \
        const float2 m_ = sR[(r & 3) + 8 * (r >> 2) + 4 * lh];                                  \
        rs[r8] = m_.x;                                                                          \
        rm[r8] = m_.y;                                                                          \
      }                                                                                         \
      _Pragma("unroll") for (int g = 0; g < NG; ++g)                                            \
        _Pragma("unroll") for (int q4 = 0; q4 < 4; ++q4) {                                      \
          float o[2];                                                                           \
          _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                       \
            const int r8 = 2 * q4 + h, r = 8 * (H) + r8;                                        \
            const float xv = acc[2 * g][r] * rs[r8] + (rm[r8] * xcs[2 * g] + xbb[2 * g]);       \
            const float gv = acc[2 * g + 1][r] * rs[r8] + (rm[r8] * xcs[2 * g + 1] + xbb[2 * g + 1]); \
            o[h] = (DBG && (p.dbg & 4)) ? xv + gv : xv * gelu_erf(gv);                          \
          }                                                                                     \
          o32[g][4 * (H) + q4] = pack_bf2_hw(o[0], o[1]);                                          \
        }                                                                                       \
    }
    PG_HALF(0)
    __builtin_amdgcn_sched_barrier(0);
    PG_HALF(1)
#undef PG_HALF
    stamp();
    // ---- stores.  The 32 x (32 NG) output patch of this wavefront is transposed through LDS so that it leaves as 16-byte stores of
    // whole rows (2-byte stores straight out of the accumulator layout cost ~130 cycles each, 4 k cycles per tile: tools/
    // pgeglu_stamps.py).  Scratch = this wavefront's OWN request pieces of the ring slot the last K step consumed: nobody reads that
    // slot any more (every wave passed the last ring wait), and the only later writer of those bytes is this wavefront's own request
    // of the next step -- no barrier needed.
    {
      constexpr int ROWB = 32 * NG * 2;                    // bytes of one patch row
      constexpr int RPPC = 1024 / ROWB;                     // patch rows per 1 KiB request piece
      constexpr int CPRW = ROWB / 16;                       // 16-byte chunks per patch row
      const int fslot = (r_slot + NST - 1 >= NST) ? r_slot - 1 : r_slot + NST - 1;
      char* const scr = smem + fslot * STAGE + wid * (8 * BK * 2);     // piece k at scr + k * RPP * BK * 2
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
          const uint16_t v = (uint16_t)((r & 1) ? (o32[g][r >> 1] >> 16) : (o32[g][r >> 1] & 0xFFFFu));
          *reinterpret_cast<uint16_t*>(scr + (rr / RPPC) * (RPP * BK * 2) + (rr % RPPC) * ROWB + (32 * g + l31) * 2) = v;
        }
      const unsigned sbase = (unsigned)(((long)(m0 + wm * 32) * p.ldc + ((n0 + wn * WTN) >> 1)) * 2);
#pragma unroll
      for (int k = 0; k < 32 * CPRW / 64; ++k) {
        const int i = k * 64 + lane, rr = i / CPRW, ch = i % CPRW;
        const u32x4 v = *reinterpret_cast<const u32x4*>(scr + (rr / RPPC) * (RPP * BK * 2) + (rr % RPPC) * ROWB + ch * 16);
        const unsigned off = (m0 + wm * 32 + rr < p.M) ? (unsigned)((rr * p.ldc + ch * 8) * 2) : OOB;
        if (!DBG || !(p.dbg & 2)) __builtin_amdgcn_raw_buffer_store_b128(v, rsC, off, off == OOB ? 0u : sbase, 16 /* sc1: write-through, see common.h st_wt */);
        else asm volatile("" ::"v"(v), "v"(off));
      }
    }
    stamp();
  }
  wait_vmcnt<0>();                 // dead-slot requests of the last steps must land before LDS is released
#undef PG_RING_SYNC
#undef PG_FRAG
#undef PG_MMA
#endif
}

template <int BM, int WGM, int WGN, int NST, int LNS, bool DBG>
hipError_t launch_pgeglu(const GemmParams& p, hipStream_t stream) {
  if (p.ln_slots > LNS || (long)p.M * p.ldc * 2 >= ((long)1 << 31)) return hipErrorInvalidValue;
  constexpr size_t lds = (size_t)NST * (BM + 128) * BK * 2 + (size_t)2 * BM * 8;
  static_assert(2 * lds <= 160 * 1024, "two resident blocks per CU");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&geglu_persistent_kernel<BM, WGM, WGN, NST, LNS, DBG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int nblk = ((p.M + BM - 1) / BM) * (p.N / 128);
  const int grid = std::min(nblk, 512);                  // 2 resident blocks on each of the 256 CUs
  hipLaunchKernelGGL((geglu_persistent_kernel<BM, WGM, WGN, NST, LNS, DBG>), dim3(grid), dim3(64 * WGM * WGN), lds, stream, p);
  return hipGetLastError();
}

}  // namespace

bool pgeglu_valid(const GemmParams& p, int tile, int batch, int splitk) {
  if ((tile == TILE_PGEGLU_128 || tile == TILE_PGEGLU_128_W8) && p.ln_slots > 10) return false;     // their row-statistics registers hold 10 slots (C <= 640)
  return p.geglu && p.ln_stats && p.ln_cs && p.bias && splitk <= 1 && batch <= 1 && p.taps == 1 && p.out_bf16 && !p.res &&
         !p.rowbias && !p.aux && !p.stats && !p.vt && p.w_rows == 0 && p.Cin2 == 0 && p.dup_rows == 0 && p.sm_w == 0 &&
         !p.relu && !p.silu && !p.store_nchw && p.alpha == 1.f && (p.N % 128) == 0 && (p.K % 64) == 0 && p.K >= 128 &&
         p.ln_slots <= 20 && p.C != nullptr;
}

hipError_t launch_gemm_pgeglu(int tile_cfg, const GemmParams& p, hipStream_t stream) {
  switch (tile_cfg) {
    case TILE_PGEGLU_128: return p.dbg ? launch_pgeglu<128, 4, 1, 2, 10, true>(p, stream) : launch_pgeglu<128, 4, 1, 2, 10, false>(p, stream);
    case TILE_PGEGLU_64: return p.dbg ? launch_pgeglu<64, 2, 2, 3, 20, true>(p, stream) : launch_pgeglu<64, 2, 2, 3, 20, false>(p, stream);
    case TILE_PGEGLU_128_W8: return p.dbg ? launch_pgeglu<128, 4, 2, 2, 10, true>(p, stream) : launch_pgeglu<128, 4, 2, 2, 10, false>(p, stream);
    case TILE_PGEGLU_128_W8L: return p.dbg ? launch_pgeglu<128, 4, 2, 2, 20, true>(p, stream) : launch_pgeglu<128, 4, 2, 2, 20, false>(p, stream);
    default: return hipErrorInvalidValue;
  }
}
