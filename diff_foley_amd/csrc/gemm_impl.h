// MFMA bf16 implicit-GEMM for gfx950: conv3x3 / conv1x1 / Linear / batched matmul with fused epilogues.
//
// Block = 256 threads = 4 wavefronts (64 lanes) in a WGM x WGN grid; each wavefront owns a
// (BM/WGM) x (BN/WGN) output tile made of 32x32 MFMA tiles (v_mfma_f32_32x32x16_bf16).
// K is walked in steps of 64.  Operand tiles go HBM -> LDS directly (buffer_load_dwordx4 ... lds, no VGPR
// round trip) into an NST-deep LDS ring: tile kt+NST-1 is requested while tile kt is multiplied, so the
// ~1 us load latency of this chip is covered even for the tiny-K GEMMs of the transformer blocks.  One raw
// s_barrier per K step; waits are counted (s_waitcnt vmcnt(N)), never a full drain inside the loop.
// LDS rows are 128 B (64 bf16); the 16-B chunk index is XOR-swizzled with (row>>1)&7 so the ds_read_b128
// fragment reads of a 16-lane group hit 16 distinct slots.  The DMA writes LDS linearly (wave base +
// lane*16), so the swizzle is applied to the per-lane SOURCE chunk instead (rule: both sides or neither).
// Out-of-range rows / conv padding use an out-of-bounds buffer offset: the hardware then writes zeros.
#include <algorithm>

#pragma once
#include "common.h"
#include "gemm.h"

namespace {

constexpr int BK = 64;

// Epilogue specialisations.  A kernel carries ONLY the epilogue it runs: the per-launch fixed cost on this chip grows by
// ~1-2 us between a 5 KB and a 45 KB code object (cold instruction fetch at every kernel boundary; tools/icache_probe.py),
// which is 10-20 % of the small GEMMs of the transformer blocks.
//   LEAN   fp32 | operand-type row-major out, bias, per-sample / per-position bias, residual   (the UNet's common case)
//   SPLITK partial slab store (the reduce kernel applies the epilogue)
//   GEGLU  value * gelu(gate) on (32 | 32) column groups, optionally LayerNorm-folded
//   PROD   LEAN + per-row partial statistics + operand-type copy (producers of a LayerNorm-folded consumer)
//   LNC    LayerNorm-folded consumer, optionally with the transposed V^T column range (fused QKV)
//   ANY    everything else: alpha, ReLU, aux copy, NCHW store, unaligned shapes (scalar fallback)
enum { EPI_LEAN = 0, EPI_SPLITK = 1, EPI_GEGLU = 2, EPI_PROD = 3, EPI_LNC = 4, EPI_ANY = 5, EPI_XS = 6 };

// Spatial patch (th x tw output pixels) owned by one block of a halo kernel with BM rows.
bool halo_patch(int H, int W, int BM, int* th, int* tw) {
  const int w = (W % 16 == 0) ? 16 : W;
  if (w <= 0 || BM % w != 0) return false;
  int h = BM / w;
  if (h > H) h = H;
  while (h > 1 && (H % h != 0 || BM % (h * w) != 0)) --h;     // tallest patch that tiles both the image and the block (192 rows: 3 x 4x16)
  if (h <= 0 || H % h != 0 || BM % (h * w) != 0) return false;
  *th = h;
  *tw = w;
  return true;
}

// Sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), result in every lane: two quad butterflies and two row
// rotations, all DPP modifiers on VALU adds -- no LDS-pipe traffic (ds_bpermute) in the epilogue.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));  // row_ror:8
  return v;
}

__device__ __forceinline__ float epi_bias(const GemmParams& p, int row, int col, float v) {
  v *= p.alpha;
  if (p.bias) v += p.bias[col];
  if (p.rowbias) {
    const int ri = (p.rowbias_mode == 1) ? (row / p.rows_per_sample) : (row % p.rows_per_sample);
    v += p.rowbias[(long)ri * p.ld_rowbias + col];
  }
  return v;
}

__device__ __forceinline__ void epi_store(const GemmParams& p, int z, int row, int col, int nout, float v) {
  if (p.silu) v = silu_f(v);
  if (p.relu) v = fmaxf(v, 0.f);
  if (p.aux) p.aux[(long)row * p.ld_aux + col] = f2bf(v);
  long idx;
  if (p.store_nchw) {
    const int b = row / p.hw_out, px = row - b * p.hw_out;
    idx = ((long)b * nout + col) * p.hw_out + px;
  } else {
    idx = (long)row * p.ldc + col;
  }
  idx += (long)z * p.c_bs;
  if (p.out_bf16)
    reinterpret_cast<bf16_t*>(p.C)[idx] = f2bf(v);
  else
    reinterpret_cast<float*>(p.C)[idx] = v;
}

__device__ __forceinline__ void epi_out(const GemmParams& p, int z, int row, int col, float v) {
  if (p.res) v += p.res[(long)z * p.res_bs + (long)row * p.ldr + col];
  epi_store(p, z, row, col, p.geglu ? (p.N >> 1) : p.N, v);
}

// Epilogue of one 32-row band of a wavefront's tile: TN 32x32 accumulator tiles side by side.
// C/D layout of the 32x32 MFMA: col = lane&31, row r -> (r&3) + 8*(r>>2) + 4*(lane>>5).  rowv[r] is the OUTPUT row
// (NHWC pixel index) of accumulator register r, or >= p.M when that row does not exist.  All loads of a tile
// (bias, per-sample bias, residual) are issued unconditionally from clamped addresses before any use, so they
// overlap instead of serialising behind per-element branches.
template <int TN>
__device__ __forceinline__ void epilogue_band(const GemmParams& p, int z, int batch, const int (&rowv)[16],
                                              f32x16 (&acc)[TN], int col0, int l31) {
  if (p.splitk > 1) {
    float* part = p.partial + (long)z * p.M * p.N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = col0 + j * 32 + l31;
      if (col >= p.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (rowv[r] < p.M) part[(long)rowv[r] * p.N + col] = acc[j][r];
    }
    return;
  }
  const bool has_bias = p.bias != nullptr, has_rb = p.rowbias != nullptr, has_res = p.res != nullptr;
  const int nout = p.geglu ? (p.N >> 1) : p.N;
  int rowc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) rowc[r] = min(rowv[r], p.M - 1);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    if (p.geglu) {
      if constexpr (TN % 2 == 0) {
        if (j & 1) continue;
        const int xcol = col0 + j * 32 + l31;
        const int xc = min(xcol, p.N - 33);
        const int ocol = (xcol >> 6) * 32 + (xcol & 63);
        const float bx = has_bias ? p.bias[xc] : 0.f, bg = has_bias ? p.bias[xc + 32] : 0.f;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float xv = acc[j][r] * p.alpha + bx;
          const float gv = acc[j + 1][r] * p.alpha + bg;
          v[r] = xv * gelu_erf(gv);
        }
        if (has_res) {
          float rr[16];
#pragma unroll
          for (int r = 0; r < 16; ++r)
            rr[r] = p.res[(long)batch * p.res_bs + (long)rowc[r] * p.ldr + min(ocol, nout - 1)];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] += rr[r];
        }
        if (xcol < p.N) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (rowv[r] < p.M) epi_store(p, batch, rowv[r], ocol, nout, v[r]);
        }
      }
      continue;
    }
    const int col = col0 + j * 32 + l31;
    const int cc = min(col, p.N - 1);
    const float b0 = has_bias ? p.bias[cc] : 0.f;
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[j][r] * p.alpha + b0;
    if (has_rb) {
      float rb_[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ri = (p.rowbias_mode == 1) ? (rowc[r] / p.rows_per_sample) : (rowc[r] % p.rows_per_sample);
        rb_[r] = p.rowbias[(long)ri * p.ld_rowbias + cc];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] += rb_[r];
    }
    if (has_res) {
      float rr[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) rr[r] = p.res[(long)batch * p.res_bs + (long)rowc[r] * p.ldr + cc];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] += rr[r];
    }
    if (col < p.N) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (rowv[r] < p.M) epi_store(p, batch, rowv[r], col, nout, v[r]);
    }
  }
}

// A thread keeps the SAME 4 columns in every step of the block epilogue's store loops (NT is a multiple of the float4 chunks
// per row), so the per-column epilogue vectors -- bias, LayerNorm column sums -- are requested ONCE, before the main loop,
// and are in registers long before the epilogue needs them (inside the loops each was an exposed L2 round trip per step).
struct EpiVec { float4 hb0, hb1, hc0, hc1; };
template <int BN, int NT, int EPI>
__device__ __forceinline__ EpiVec epi_prefetch(const GemmParams& p, int n0, int tid) {
  EpiVec ev;
  ev.hb0 = ev.hb1 = ev.hc0 = ev.hc1 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.store_nchw || (p.N & 3)) return ev;      // scalar epilogue (epilogue_band) reads its own
  if constexpr (EPI == EPI_GEGLU) {
    static_assert(NT % (BN / 8) == 0, "invariant column per thread");
    const int oc = (tid % (BN / 8)) * 4, xc = (oc >> 5) * 64 + (oc & 31);
    const int gx = min(n0 + xc, p.N - 36);
    if (p.ln_stats) {
      ev.hc0 = *reinterpret_cast<const float4*>(&p.ln_cs[gx]);
      ev.hc1 = *reinterpret_cast<const float4*>(&p.ln_cs[gx + 32]);
    }
    if (p.bias) {
      ev.hb0 = *reinterpret_cast<const float4*>(&p.bias[gx]);
      ev.hb1 = *reinterpret_cast<const float4*>(&p.bias[gx + 32]);
    }
  } else if constexpr (EPI != EPI_SPLITK && EPI != EPI_XS) {
    static_assert(NT % (BN / 4) == 0, "invariant column per thread");
    const int cch = min(n0 + (tid % (BN / 4)) * 4, p.N - 4);
    if (p.bias) ev.hb0 = *reinterpret_cast<const float4*>(&p.bias[cch]);
    if constexpr (EPI == EPI_LNC)
      if (p.ln_stats) ev.hc0 = *reinterpret_cast<const float4*>(&p.ln_cs[cch]);
  }
  return ev;
}

// Block-level epilogue through LDS: the accumulators of the whole BM x BN tile are parked in LDS as fp32 (row
// stride BN+4 floats keeps the 16-B reads conflict-free), then every thread re-reads 4 consecutive columns of one
// row, applies alpha / bias / per-sample bias / residual / GEGLU on float4s and issues ONE 16-byte (fp32) or 8-byte
// (bf16) store: rows leave the CU as full 128..512-B contiguous runs instead of 4-B-per-lane column slivers.
// ROWMAP(r) gives the output row (NHWC pixel index) of tile row r, or >= p.M when the row does not exist.
// Requires N % 4 == 0, ldc % 4 == 0 and row-major output (the caller falls back to epilogue_band otherwise).
// What a wavefront that takes no part in epilogue_block has to execute so that the block's barriers still count it:
// exactly epilogue_block's barriers (keep in step with it).
__device__ __forceinline__ void epilogue_block_idle() {
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_s_barrier();
}
template <int BM, int BN, int NT, int TM, int TN, int EPI, bool PARK = true, class RowMap>
__device__ __forceinline__ void epilogue_block(const GemmParams& p, int z, int batch, float* sC,
                                               f32x16 (&acc)[TM][TN], int wrow0, int wcol0, int n0, int tid,
                                               RowMap rowmap, const EpiVec& ev, float2 ln_mr = make_float2(0.f, 1.f)) {
  constexpr int LDC = BN + 4;
  const int lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  float2* sRow = reinterpret_cast<float2*>(sC + BM * LDC);   // (mean, rstd) of tile row r (LayerNorm-folded GEMMs)
  const float4 hb0 = ev.hb0, hb1 = ev.hb1, hc0 = ev.hc0, hc1 = ev.hc1;
#define DF_EPI_STAMP(K) \
  if ((p.dbg & 64) && p.splitk <= 1 && p.partial && tid == 0) \
    reinterpret_cast<unsigned long long*>(p.partial)[(long)blockIdx.x * 32 + 24 + (K)] = __builtin_amdgcn_s_memtime();
  DF_EPI_STAMP(0)
  __builtin_amdgcn_s_barrier();            // every wave is done reading the operand ring
  if ((EPI == EPI_LNC || EPI == EPI_GEGLU || EPI == EPI_XS) && p.ln_stats && tid < BM) sRow[tid] = ln_mr;
  if constexpr (PARK) {        // (producer wavefronts of a producer-specialised block hold no accumulators: they join at the store loops)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sC[(wrow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDC + wcol0 + j * 32 + l31] = acc[i][j][r];
  }
  __syncthreads();
  DF_EPI_STAMP(1)
  const bool has_bias = p.bias != nullptr, has_rb = p.rowbias != nullptr, has_res = p.res != nullptr;
  if constexpr (EPI == EPI_SPLITK) {
    float* part = p.partial + (long)z * p.M * p.N;
    constexpr int CPR = BN / 4;
#pragma unroll 4
    for (int e = tid; e < BM * CPR; e += NT) {
      const int r = e / CPR, c4 = (e - r * CPR) * 4;
      const int row = rowmap(r), col = n0 + c4;
      const float4 v = *reinterpret_cast<const float4*>(&sC[r * LDC + c4]);
      if (row < p.M && col < p.N) st_wt(reinterpret_cast<float4*>(&part[(long)row * p.N + col]), v);
    }
    return;
  } else if constexpr (EPI == EPI_XS) {
    // cross-attention scores: LayerNorm fold with this sample's column sums / bias, then softmax over every 32-column head
    // group (8 consecutive lanes hold one group of one row), operand-type probabilities out
    constexpr int CPR = BN / 4;
#pragma unroll 2
    for (int ei = 0; ei < BM * CPR / NT; ++ei) {
      const int e = tid + ei * NT;
      const int r = e / CPR, c4 = (e - r * CPR) * 4;
      const int row = rowmap(r), col = n0 + c4;
      const int rc = min(row, p.M - 1), cc = min(col, p.N - 4);
      float4 v = *reinterpret_cast<const float4*>(&sC[r * LDC + c4]);
      const float2 mr = sRow[r];
      const long so = (long)(rc / p.w_rows) * p.N + cc;
      const float4 cs = *reinterpret_cast<const float4*>(&p.ln_cs[so]);
      const float4 bb = *reinterpret_cast<const float4*>(&p.bias[so]);
      v.x = mr.y * (v.x - mr.x * cs.x) + bb.x; v.y = mr.y * (v.y - mr.x * cs.y) + bb.y;
      v.z = mr.y * (v.z - mr.x * cs.z) + bb.z; v.w = mr.y * (v.w - mr.x * cs.w) + bb.w;
      const int g0 = col & 31;
      const float NEG = -3.0e38f;
      if (g0 + 0 >= p.sm_valid) v.x = NEG;
      if (g0 + 1 >= p.sm_valid) v.y = NEG;
      if (g0 + 2 >= p.sm_valid) v.z = NEG;
      if (g0 + 3 >= p.sm_valid) v.w = NEG;
      float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
      mx = fmaxf(mx, dpp_xor1(mx));             // the 8 lanes that hold a head's 32 columns of this row
      mx = fmaxf(mx, dpp_xor2(mx));
      mx = fmaxf(mx, dpp_half_mirror(mx));
      v.x = g0 + 0 < p.sm_valid ? __expf(v.x - mx) : 0.f;
      v.y = g0 + 1 < p.sm_valid ? __expf(v.y - mx) : 0.f;
      v.z = g0 + 2 < p.sm_valid ? __expf(v.z - mx) : 0.f;
      v.w = g0 + 3 < p.sm_valid ? __expf(v.w - mx) : 0.f;
      float sm = (v.x + v.y) + (v.z + v.w);
      sm += dpp_xor1(sm);
      sm += dpp_xor2(sm);
      sm += dpp_half_mirror(sm);
      const float inv = 1.0f / sm;
      if (row < p.M && col < p.N)
        st_wt(reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + (long)row * p.ldc + col),
              make_uint2(pack_bf2(v.x * inv, v.y * inv), pack_bf2(v.z * inv, v.w * inv)));
    }
    return;
  } else if constexpr (EPI == EPI_GEGLU) {
    constexpr int CPR = BN / 8;              // float4 chunks of OUTPUT columns per row (output width BN/2)
    const int nout = p.N >> 1;
#pragma unroll 2
    for (int e = tid; e < BM * CPR; e += NT) {
      const int r = e / CPR, oc = (e - r * CPR) * 4;           // output column within the tile
      const int xc = (oc >> 5) * 64 + (oc & 31);               // packed x column within the tile; gate at +32
      const int row = rowmap(r), ocol = (n0 >> 1) + oc;
      float4 x = *reinterpret_cast<const float4*>(&sC[r * LDC + xc]);
      float4 g = *reinterpret_cast<const float4*>(&sC[r * LDC + xc + 32]);
      if (p.ln_stats) {       // LayerNorm folded in: rstd * (acc - mean * colsum(gamma W))
        const float2 mr = sRow[r];
        const float4 cx = hc0, cg = hc1;
        x.x = mr.y * (x.x - mr.x * cx.x); x.y = mr.y * (x.y - mr.x * cx.y); x.z = mr.y * (x.z - mr.x * cx.z); x.w = mr.y * (x.w - mr.x * cx.w);
        g.x = mr.y * (g.x - mr.x * cg.x); g.y = mr.y * (g.y - mr.x * cg.y); g.z = mr.y * (g.z - mr.x * cg.z); g.w = mr.y * (g.w - mr.x * cg.w);
      }
      if (has_bias) {
        const float4 bx = hb0, bg = hb1;
        x.x = x.x * p.alpha + bx.x; x.y = x.y * p.alpha + bx.y; x.z = x.z * p.alpha + bx.z; x.w = x.w * p.alpha + bx.w;
        g.x = g.x * p.alpha + bg.x; g.y = g.y * p.alpha + bg.y; g.z = g.z * p.alpha + bg.z; g.w = g.w * p.alpha + bg.w;
      }
      float4 v = make_float4(x.x * gelu_erf(g.x), x.y * gelu_erf(g.y), x.z * gelu_erf(g.z), x.w * gelu_erf(g.w));
      if (row < p.M && ocol < nout) {
        const long idx = (long)batch * p.c_bs + (long)row * p.ldc + ocol;
        if (p.out_bf16)
          st_wt(reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + idx), make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)));
        else
          st_wt(reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + idx), v);
      }
    }
    return;
  } else {
  constexpr int CPR = BN / 4;
  // rows_per_sample is a power of two for every latent this model is used with: shift instead of a 25-instruction divide
  const int rps_sh = (has_rb && p.rows_per_sample > 0 && (p.rows_per_sample & (p.rows_per_sample - 1)) == 0)
                         ? (31 - __builtin_clz(p.rows_per_sample)) : -1;
  // ---- fused QKV projection: the V columns leave transposed, V^T[sample][col][token], 4 tokens (8 B) per store
  if (EPI == EPI_LNC && p.vt && n0 >= p.vt_col0) {
    constexpr int RG = BM / 4, CG = BN / 4;   // 4-row (token) groups x 4-column groups: a 4x4 block per thread step
    const int cv = p.N - p.vt_col0;
#pragma unroll 2
    for (int e = tid; e < CG * RG; e += NT) {
      const int cg = e / RG, r = (e - cg * RG) * 4, c4 = cg * 4;   // lanes run along the tokens of one column group
      const int row = rowmap(r), col = n0 + c4;
      const int cc = min(col, p.N - 4);
      float4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = *reinterpret_cast<const float4*>(&sC[(r + j) * LDC + c4]);
        v[j].x *= p.alpha; v[j].y *= p.alpha; v[j].z *= p.alpha; v[j].w *= p.alpha;
      }
      if (p.ln_stats) {
        const float4 cs = *reinterpret_cast<const float4*>(&p.ln_cs[cc]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float2 mr = sRow[r + j];
          v[j].x = mr.y * (v[j].x - mr.x * cs.x); v[j].y = mr.y * (v[j].y - mr.x * cs.y);
          v[j].z = mr.y * (v[j].z - mr.x * cs.z); v[j].w = mr.y * (v[j].w - mr.x * cs.w);
        }
      }
      if (has_bias) {
        const float4 b = *reinterpret_cast<const float4*>(&p.bias[cc]);
        // The .y column of two rows is one packed add of (b.y, b.y) -- the HIGH register of the (b.x, b.y) pair for BOTH result
        // lanes.  The compiler encodes that with the operand select on src1 (`v_pk_add_f32 D, D, B op_sel:[0,1]`), which reads the
        // operand as 0 in lanes 48-63 while another wavefront of the SIMD executes MFMAs (common.h pk_add_hi; the fused QKV
        // projection's V^T rows lost their bias at random under two blocks per CU).  Written out in the src0 form instead.
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j].x += b.x; v[j].z += b.z; v[j].w += b.w; }
        pk_add_hi(v[0].y, v[1].y, b.x, b.y);
        pk_add_hi(v[2].y, v[3].y, b.x, b.y);
      }
      if (row < p.M && col < p.N) {           // M % 4 == 0 and vt_T % 4 == 0: the 4 rows belong to one sample
        const int smp = row / p.vt_T, t = row - smp * p.vt_T;
        bf16_t* dst = p.vt + ((long)smp * cv + (col - p.vt_col0)) * p.ldvt + t;
        st_wt(reinterpret_cast<uint2*>(dst), make_uint2(pack_bf2(v[0].x, v[1].x), pack_bf2(v[2].x, v[3].x)));
        st_wt(reinterpret_cast<uint2*>(dst + p.ldvt), make_uint2(pack_bf2(v[0].y, v[1].y), pack_bf2(v[2].y, v[3].y)));
        st_wt(reinterpret_cast<uint2*>(dst + 2 * p.ldvt), make_uint2(pack_bf2(v[0].z, v[1].z), pack_bf2(v[2].z, v[3].z)));
        st_wt(reinterpret_cast<uint2*>(dst + 3 * p.ldvt), make_uint2(pack_bf2(v[0].w, v[1].w), pack_bf2(v[2].w, v[3].w)));
      }
    }
    return;
  }
  // FL bit 0: ReLU and/or a second (operand-type) copy of the stored value; bit 1: LayerNorm folded into this GEMM;
  // bit 2: per-row partial statistics of the stored value (16 lanes = one 64-column slot of one row).  Every variant
  // is instantiated separately so the UNet's plain epilogues keep their lean loop body.
#define DF_EPI_LOOP(FL)                                                                                             \
  _Pragma("unroll 4") for (int ei = 0; ei < BM * CPR / NT; ++ei) {                                                  \
    const int e = tid + ei * NT;                                                                                    \
    const int r = e / CPR, c4 = (e - r * CPR) * 4;                                                                  \
    const int row = rowmap(r), col = n0 + c4;                                                                       \
    const int rc = min(row, p.M - 1), cc = min(col, p.N - 4);                                                       \
    float4 v = *reinterpret_cast<const float4*>(&sC[r * LDC + c4]);                                                 \
    if ((FL) & 16) { v.x *= p.alpha; v.y *= p.alpha; v.z *= p.alpha; v.w *= p.alpha; }                              \
    if ((FL) & 2) {                                                                                                 \
      const float2 mr = sRow[r];                                                                                    \
      const float4 cs = hc0;                                                                                        \
      v.x = mr.y * (v.x - mr.x * cs.x); v.y = mr.y * (v.y - mr.x * cs.y);                                           \
      v.z = mr.y * (v.z - mr.x * cs.z); v.w = mr.y * (v.w - mr.x * cs.w);                                           \
    }                                                                                                               \
    if (has_bias_l) {                                                                                               \
      const float4 b = hb0;                                                                                         \
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;                                                               \
    }                                                                                                               \
    if (has_rb_l) {                                                                                                 \
      const int ri = (rps_sh >= 0) ? ((p.rowbias_mode == 1) ? (rc >> rps_sh) : (rc & (p.rows_per_sample - 1)))      \
                                   : ((p.rowbias_mode == 1) ? (rc / p.rows_per_sample) : (rc % p.rows_per_sample)); \
      const float4 b = *reinterpret_cast<const float4*>(&p.rowbias[(long)ri * p.ld_rowbias + cc]);                  \
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;                                                               \
    }                                                                                                               \
    if (has_res_l) {                                                                                                \
      const float4 b = *reinterpret_cast<const float4*>(&p.res[(long)batch * p.res_bs + (long)rc * p.ldr + cc]);    \
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;                                                               \
    }                                                                                                               \
    if (!((FL) & (2 | 4)) && p.silu) {                                                                              \
      v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w);                                   \
    }                                                                                                               \
    if (((FL) & 1) && p.relu) {                                                                                     \
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);                   \
    }                                                                                                               \
    const bool ok = row < p.M && col < p.N;                                                                         \
    if ((FL) & 4) {                                                                                                 \
      const float s1 = row16_sum(ok ? (v.x + v.y) + (v.z + v.w) : 0.f);                                             \
      const float s2 = row16_sum(ok ? (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w) : 0.f);                     \
      if (ok && (tid & 15) == 0) {                                                                                  \
        st_wt(&p.stats[(long)row * p.stats_slots + (col >> 6)], make_float2(s1, s2));                                      \
        if (p.dup_rows) st_wt(&p.stats[(long)(row + p.dup_rows) * p.stats_slots + (col >> 6)], make_float2(s1, s2));       \
      }                                                                                                             \
    }                                                                                                               \
    if (ok) {                                                                                                       \
      const long idx = (long)batch * p.c_bs + (long)row * p.ldc + col;                                              \
      if (((FL) & 4) && p.no_c_store) { /* producer whose fp32 value nobody reads: operand copy + statistics only */ \
      } else if (p.out_bf16)                                                                                        \
        st_wt(reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + idx), make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w))); \
      else                                                                                                          \
        st_wt(reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + idx), v);                                        \
      if (((FL) & 1) && p.aux)                                                                                      \
        st_wt(reinterpret_cast<uint2*>(p.aux + (long)row * p.ld_aux + col), make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w))); \
      if (p.dup_rows) {       /* CFG prefix: the other half of the batch gets the same row */                       \
        const long idx2 = idx + (long)p.dup_rows * p.ldc;                                                           \
        if (((FL) & 4) && p.no_c_store) {                                                                            \
        } else if (p.out_bf16)                                                                                      \
          st_wt(reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + idx2), make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w))); \
        else                                                                                                        \
          st_wt(reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + idx2), v);                                     \
        if (((FL) & 1) && p.aux)                                                                                    \
          st_wt(reinterpret_cast<uint2*>(p.aux + (long)(row + p.dup_rows) * p.ld_aux + col), make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w))); \
      }                                                                                                             \
    }                                                                                                               \
  }
  // Per-sample bias (FiLM) and fp32 residual of the plain epilogues: ALL their loads first, then added into the parked tile.
  // Inside the store loop each of them was `load; s_waitcnt vmcnt(0)` per iteration (run-time-optional loads sit behind branches,
  // and the residual may alias C, so nothing could be hoisted over the previous iteration's store): a chain of up to 2 x 16
  // round trips, 6.9 k of the 52 k cycles of a full-resolution conv (tools/halo_stamps.py).  Same summation order as before
  // ((acc + bias) + FiLM) + residual, so results are bit-identical; every thread touches only the elements it stores later.
  bool pre_added = false;
  if constexpr (EPI == EPI_PROD || EPI == EPI_LEAN) {
    constexpr int ITERS = BM * CPR / NT;
    if (has_rb || has_res) {
      float4 rbv[ITERS], rsv[ITERS];
      if (has_rb) {
#pragma unroll
        for (int ei = 0; ei < ITERS; ++ei) {
          const int e = tid + ei * NT, r = e / CPR, c4 = (e - r * CPR) * 4;
          const int rc = min(rowmap(r), p.M - 1), cc = min(n0 + c4, p.N - 4);
          const int ri = (rps_sh >= 0) ? ((p.rowbias_mode == 1) ? (rc >> rps_sh) : (rc & (p.rows_per_sample - 1)))
                                       : ((p.rowbias_mode == 1) ? (rc / p.rows_per_sample) : (rc % p.rows_per_sample));
          rbv[ei] = *reinterpret_cast<const float4*>(&p.rowbias[(long)ri * p.ld_rowbias + cc]);
        }
      }
      if (has_res) {
#pragma unroll
        for (int ei = 0; ei < ITERS; ++ei) {
          const int e = tid + ei * NT, r = e / CPR, c4 = (e - r * CPR) * 4;
          const int rc = min(rowmap(r), p.M - 1), cc = min(n0 + c4, p.N - 4);
          rsv[ei] = *reinterpret_cast<const float4*>(&p.res[(long)batch * p.res_bs + (long)rc * p.ldr + cc]);
        }
      }
#pragma unroll
      for (int ei = 0; ei < ITERS; ++ei) {
        const int e = tid + ei * NT, r = e / CPR, c4 = (e - r * CPR) * 4;
        float4 v = *reinterpret_cast<const float4*>(&sC[r * LDC + c4]);
        if (has_bias) { v.x += hb0.x; v.y += hb0.y; v.z += hb0.z; v.w += hb0.w; }
        if (has_rb) { v.x += rbv[ei].x; v.y += rbv[ei].y; v.z += rbv[ei].z; v.w += rbv[ei].w; }
        if (has_res) { v.x += rsv[ei].x; v.y += rsv[ei].y; v.z += rsv[ei].z; v.w += rsv[ei].w; }
        *reinterpret_cast<float4*>(&sC[r * LDC + c4]) = v;
      }
      pre_added = true;
    }
  }
  DF_EPI_STAMP(2)
  const bool has_bias_l = has_bias && !pre_added, has_rb_l = has_rb && !pre_added, has_res_l = has_res && !pre_added;
  // Lean store loops for the two common cases.  The general loop below decides everything per iteration (output type, second
  // copy, duplicated rows, activations, 64-bit index products): ~90 instruction slots = 375 cycles per 16-byte store, 4.5 k cycles
  // for the 12 stores of a full-resolution conv tile (tools/halo_stamps.py).  A thread keeps one column chunk and walks rows at a
  // fixed stride, so here an iteration is: tile row -> output row, one LDS read, (bias), store.
  if constexpr (EPI == EPI_LEAN || EPI == EPI_PROD) {
    constexpr int ITERS = BM * CPR / NT, RSTEP = NT / CPR;
    const bool plain = !p.dup_rows && !p.silu && !p.relu && (EPI == EPI_PROD || (!p.out_bf16 && !p.aux));
    if (plain) {
      const int rb0 = tid / CPR, c4 = (tid - rb0 * CPR) * 4, col = n0 + c4;
      const bool colok = col < p.N;
      const long cbase = (long)batch * p.c_bs + col;
#pragma unroll
      for (int ei = 0; ei < ITERS; ++ei) {
        const int r = rb0 + ei * RSTEP;
        const int row = rowmap(r);
        float4 v = *reinterpret_cast<const float4*>(&sC[r * LDC + c4]);
        if (has_bias_l) { v.x += hb0.x; v.y += hb0.y; v.z += hb0.z; v.w += hb0.w; }
        const bool ok = colok && row < p.M;
        if constexpr (EPI == EPI_PROD) {
          const float s1 = row16_sum(ok ? (v.x + v.y) + (v.z + v.w) : 0.f);
          const float s2 = row16_sum(ok ? (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w) : 0.f);
          if (ok) {
            if ((tid & 15) == 0) st_wt(&p.stats[(long)row * p.stats_slots + (col >> 6)], make_float2(s1, s2));
            if (!p.no_c_store) {
              if (p.out_bf16)
                st_wt(reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + cbase + (long)row * p.ldc), make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)));
              else
                st_wt(reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + cbase + (long)row * p.ldc), v);
            }
            if (p.aux) st_wt(reinterpret_cast<uint2*>(p.aux + (long)row * p.ld_aux + col), make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)));
          }
        } else {
          if (ok) st_wt(reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + cbase + (long)row * p.ldc), v);
        }
      }
      return;
    }
  }
  if constexpr (EPI == EPI_PROD) DF_EPI_LOOP(5)
  else if constexpr (EPI == EPI_LNC) DF_EPI_LOOP(2)
  else if constexpr (EPI == EPI_ANY) {
    if (p.relu || p.aux) DF_EPI_LOOP(17)
    else DF_EPI_LOOP(16)
  } else DF_EPI_LOOP(0)
#undef DF_EPI_LOOP
  }
}

// tools (tools/race_hunt.py): -DDF_HUNT=1 drains every LDS-DMA request at each K-step boundary of the symmetric generic kernel
#if defined(DF_HUNT) && (DF_HUNT & 1)
#define DF_HUNT_VM(VM) 0
#else
#define DF_HUNT_VM(VM) (VM)
#endif
template <int N_>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

// Quotient of small non-negative integers by a block-uniform divisor, 4 instructions instead of the ~30 of the general 32-bit
// division: q = trunc((x + 0.5) * (1/d)) is exact for 0 <= x < 2^22 (the product's error (q+1) 2^-23 stays below the 0.5/d
// distance of (x + 0.5)/d from the next integer).  The kernels' prologues turn tile rows into (sample, y, x) with up to 60 such
// divisions per thread before the first operand request can be issued -- ~3 us of the ~4.4 us launch floor of a halo conv.
struct FastDiv {
  float inv;
  int d;
  __device__ __forceinline__ explicit FastDiv(int d_) : inv(1.0f / (float)d_), d(d_) {}
  __device__ __forceinline__ int div(int x) const { return (int)(((float)x + 0.5f) * inv); }
};

// Logical tile id -> (M tile, N tile): M-fastest inside row groups of `gm` M-tiles (gm <= 0: one group = plain M-fastest).
__device__ __forceinline__ void tile_of(int lid, int nbm, int nbn, int gm, int& mt, int& nt) {
  if (gm <= 0 || gm >= nbm) {
    mt = lid % nbm;
    nt = lid / nbm;
    return;
  }
  const int per = gm * nbn, grp = lid / per, rem = lid - grp * per;
  const int rows = min(gm, nbm - grp * gm);
  nt = rem / rows;
  mt = grp * gm + (rem - nt * rows);
}

// NSTB = depth of the W ring when it differs from the A ring's NST ("deep weight ring" tiles for the weight-streaming layers):
// the weight tiles are requested NSTB - 1 K steps ahead, the activation tiles NST - 1.  Why: M <= 512 layers stream 26-59 MB of
// weights that arrive COLD from HBM every step (1.7 GB of weights per step against 256 MB of Infinity Cache), and every weight
// tile is requested by all M tiles of the layer at once, so the UNIQUE bytes in flight are (blocks x W bytes in flight per
// block) / (M tiles).  With one 16 KB stage per block that is ~1.5 MB chip-wide = 0.8 TB/s at ~2 us of loaded HBM latency --
// the rate these layers were measured at.  Activations are L2 hits (just written, re-read by every N tile): two stages suffice.
//
// PS = 1 (round 4): PRODUCER-SPECIALISED block.  The block carries a second set of WGM x WGN wavefronts that do nothing but issue
// the LDS-DMA requests of the operand ring; the first set ("consumers") only reads fragments and runs MFMAs.  Motivation
// (tools/halo_stamps.py, tools/pgeglu_stamps.py): an LDS-DMA request blocks the issuing wavefront for ~100 cycles and a
// wavefront issues in order, so in the symmetric kernel a K step costs (requests per wave x ~100) + (MFMAs per wave x 32)
// cycles -- the "17.5 B/clk per block" delivery wall every tile shape, ring depth and wave count showed.  Two co-resident
// blocks overlap the two phases by accident of phase (30 B/clk); here the overlap is by construction: each SIMD holds one
// consumer and one producer wavefront, the SIMD issues the producer's VMEM requests while the consumer's MFMAs run.  Both sets
// meet at the same one s_barrier per K step; the ring protocol (slot of tile it-1 refilled during tile it) is unchanged.
// PS = 2: two producer wavefronts per consumer (12 wavefronts): a wavefront sustains one request per ~130 cycles, 4 producers
// deliver ~31 B/clk (tools/gemm_bench.py: 128 x 128 tile, 1030 cycles per K step against 512 of MFMA).
// After the K loop the first WGM x WGN producers join the consumers in the epilogue's store loops (2 * NT threads); further
// producers only keep the epilogue's barriers company.
template <int BM, int BN, int WGM, int WGN, int NST, int MODE, int EPI, int NSTB, int PS>
__device__ __forceinline__ void gemm_bf16_body(const GemmParams& p);
template <int BM, int BN, int WGM, int WGN, int NST, int MODE, int EPI, int NSTB = NST, int PS = 0>
__global__ __launch_bounds__(64 * WGM * WGN * (1 + PS)) void gemm_bf16_kernel(GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)   // host pass only needs the launch stub (LDS-DMA builtins do not parse there)
  const DfTouch ka = gemm_kernarg_touch();      // the kernel-argument lines into L2, beside the first scalar loads (gemm.h)
  gemm_bf16_body<BM, BN, WGM, WGN, NST, MODE, EPI, NSTB, PS>(p);
  gemm_kernarg_touch_end(ka);
#endif
}
template <int BM, int BN, int WGM, int WGN, int NST, int MODE, int EPI, int NSTB, int PS>
__device__ __forceinline__ void gemm_bf16_body(const GemmParams& p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int NT = 64 * WGM * WGN;          // threads (4 or 8 wavefronts); PS: the CONSUMER threads
  constexpr int NTP = PS ? NT * PS : NT;      // threads that issue the DMA requests (PS: PS producer wavefronts per consumer wavefront)
  constexpr int RPP = NTP / 8;                // LDS rows filled per DMA pass of the block
  constexpr int AP = BM / RPP, BP = BN / RPP;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the DMA pass");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(NST >= 2 && NST <= 5, "ring depth");
  static_assert(NSTB >= NST && NSTB <= 16, "weight ring depth");
  constexpr int WD = NSTB - NST;              // extra K steps the weight requests run ahead of the activation requests
  constexpr int WSLOT = BN * BK * 2;          // bytes of one weight ring slot

  static_assert(PS == 0 || NSTB == NST, "producer-specialised blocks use the shared ring");
  const int raw_wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const bool producer = PS && raw_wid >= WGM * WGN;            // wave-uniform role
  // role-local thread / wave index: the staging coordinates of a producer are those the same thread index has in the symmetric kernel
  const int tid = producer ? (int)threadIdx.x - NT : (int)threadIdx.x, lane = tid & 63, wid = producer ? raw_wid - WGM * WGN : raw_wid;
  const int wm = wid / WGN, wn = wid % WGN;
  const int l31 = lane & 31, lh = lane >> 5;

  // tools (p.dbg bit 6, split-K 1): thread 0 stamps the shader clock at phase boundaries into p.partial[block][32] (uint64):
  // entry, operand requests of the prologue issued, first tile landed, K loop done; the epilogue's stamps follow at slots 24..
  int n_stamp = 0;
  const unsigned long long t_entry = __builtin_amdgcn_s_memtime();     // before any kernel argument is needed (slot 31 of the stamp row)
  auto stamp = [&]() {
    if ((p.dbg & 64) && p.splitk <= 1 && p.partial && threadIdx.x == 0 && blockIdx.z == 0 && n_stamp < 24) {
      if (n_stamp == 0) reinterpret_cast<unsigned long long*>(p.partial)[(long)blockIdx.x * 32 + 31] = t_entry;
      reinterpret_cast<unsigned long long*>(p.partial)[(long)blockIdx.x * 32 + n_stamp++] = __builtin_amdgcn_s_memtime();
    }
  };
  stamp();
  // ---- tile id with XCD-aware remap (block b runs on XCD b%8; give each XCD a contiguous range)
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int nblk = nbm * nbn;
  int lid;
  {
    const int bid = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // each XCD owns a contiguous range of logical tiles, walked M-fastest inside row groups of p.gm M-tiles: the walk
  // decides which A / W panels the XCD's 4 MB L2 gets to share (a miss is served by MALL at ~1/5 of the L2 rate)
  int mt_, nt_;
  tile_of(lid, nbm, nbn, p.gm, mt_, nt_);
  const int m0 = mt_ * BM, n0 = nt_ * BN;

  // LayerNorm folded into this GEMM: thread r < BM requests the producer's per-row (sum, sumsq) partials of tile row r FIRST
  // (<= LNS 8-byte loads, in flight under the operand prologue) and folds them to (mean, rstd) just before the first ring wait
  // -- by then they have landed (loads retire in order and these are older than the operand requests).  Two registers per thread
  // ride through the main loop; any tile shape can serve the 1280-channel level (20 slots per row), and nothing is exchanged
  // through LDS after the loop.  (Rounds 2-3 carried <= 5 thread-linear float2 per thread and reduced after the loop: 256-row
  // tiles at C = 1280 did not fit, and the exchange cost a barrier + an LDS round trip per block.)
  constexpr int LNS = 20;
  const bool ln_on = MODE == 0 && (EPI == EPI_LNC || EPI == EPI_GEGLU || EPI == EPI_XS) && p.ln_stats != nullptr;
  float2 lnv[LNS];
  if (ln_on && !producer) {
    const float2* sp = p.ln_stats + (long)min(m0 + min(tid, BM - 1), p.M - 1) * p.ln_slots;
#pragma unroll
    for (int i = 0; i < LNS; ++i)
      if (i < p.ln_slots) lnv[i] = sp[i];
  }

  int z = blockIdx.z;
  // MODE 3 (nearest-x2 upsample + conv3x3 as four 2x2-tap convs, one per output phase (a, b) = (Y & 1, X & 1)): grid z =
  // phase + 4 * split, one weight matrix per phase
  const int phase = (MODE == 3) ? (z & 3) : 0;
  if (MODE == 3) z >>= 2;
  const int nk = p.K / BK;
  int kt0 = 0, kt1 = nk, batch = (MODE == 3) ? phase : z;
  if (p.splitk > 1) {
    const int per = (nk + p.splitk - 1) / p.splitk;
    kt0 = z * per;
    kt1 = min(nk, kt0 + per);
    if (MODE != 3) batch = 0;
  }
  const bf16_t* Ab = p.A + (long)batch * p.a_bs;
  const bf16_t* Wb = p.W + (long)batch * p.w_bs + (p.w_rows > 0 ? (long)(m0 / p.w_rows) * p.w_bs : 0l);

  // ---- per-thread staging coordinates: chunk c (8 bf16 = 16 B) of rows (tid>>3) + 32*i.
  // Operands are fetched with raw buffer loads: an out-of-range byte offset (OOB) makes the hardware return
  // zeros, so zero padding / ragged tiles need no branches.  Everything the inner loop needs is hoisted: a DMA
  // request costs one v_add (+ one v_cndmask for conv padding), an LDS fragment read costs no VALU at all
  // (per-k-step byte offsets are precomputed, the ring slot is a compile-time immediate).
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, (int)p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(MODE <= 1 && p.A2 ? p.A2 : Ab), 0,
                                                                          (int)(MODE <= 1 && p.A2 ? p.a2_bytes : 0u), 0x00020000);
  // MODE 0 with a second A tensor: K columns [K - Cin2, K) come from A2 [M][lda2] (two linear layers merged into one GEMM)
  const unsigned k1b = (MODE == 0 && p.Cin2 > 0) ? (unsigned)(p.K - p.Cin2) * 2u : 0xffffffffu;
  const int r0 = tid >> 3;
  const int c8 = ((tid & 7) ^ ((r0 >> 1) & 7)) * 8;   // source chunk that lands in LDS slot (tid&7) of row r0+32i
  unsigned a_off[AP];   // MODE 0: byte offset of (row, chunk); MODE 1: byte offset of the centre-tap pixel;
                        // MODE 2: pixel index base n*H*W
  unsigned a_msk[AP];   // MODE 1: bit t set <=> tap t of this row reads inside the image
  unsigned a_off2[AP];  // MODE 1 with a folded skip connection: byte offset of the centre pixel in the second tensor
  int a_iy[AP], a_ix[AP];
  const int UH = p.H << p.ups, UW = p.Wd << p.ups;
  const bool fast_rows = MODE != 0 && p.M + BM < (1 << 22);      // row -> (sample, y, x) by FastDiv (exact below 2^22)
  const FastDiv fd_ohw(MODE != 0 ? max(p.OH * p.OW, 1) : 1), fd_ow(MODE != 0 ? max(p.OW, 1) : 1);
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int m = m0 + r0 + RPP * i;
    const bool mv = m < p.M;
    a_msk[i] = 0;
    a_off2[i] = OOB;
    a_iy[i] = a_ix[i] = 0;
    if (MODE == 0) {
      a_off[i] = mv ? (unsigned)(((long)m * p.lda + c8) * 2) : OOB;
      if (p.Cin2 > 0) a_off2[i] = mv ? (unsigned)(((long)m * p.lda2 + c8) * 2) : OOB;
    } else {
      const int ohw = p.OH * p.OW;
      int nb, rem, oy;
      if (fast_rows) {
        nb = fd_ohw.div(m); rem = m - nb * ohw;
        oy = fd_ow.div(rem);
      } else {
        nb = m / ohw; rem = m - nb * ohw;
        oy = rem / p.OW;
      }
      const int ox = rem - oy * p.OW;
      if (MODE == 3) {
        // rows = INPUT-resolution pixels (OH = H, OW = Wd here); tap (dy, dx) of phase (a, b) reads pixel (y-1+a+dy, x-1+b+dx)
        const int pa = phase >> 1, pb = phase & 1;
        a_off[i] = (unsigned)((((long)(nb * p.H + oy - 1 + pa) * p.Wd + ox - 1 + pb) * p.lda + c8) * 2);
        unsigned msk = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int y = oy - 1 + pa + (t >> 1), x = ox - 1 + pb + (t & 1);
          if (mv && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.Wd) msk |= 1u << t;
        }
        a_msk[i] = msk;
      } else if (MODE == 1) {
        a_off2[i] = mv ? (unsigned)((((long)(nb * p.H + oy) * p.Wd + ox) * p.lda2 + c8) * 2) : OOB;
        a_off[i] = (unsigned)((((long)(nb * p.H + oy) * p.Wd + ox) * p.lda + c8) * 2);
        unsigned msk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int y = oy + t / 3 - 1, x = ox + t % 3 - 1;
          if (mv && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.Wd) msk |= 1u << t;
        }
        a_msk[i] = msk;
      } else {
        a_off[i] = (unsigned)(nb * p.H * p.Wd);
        a_iy[i] = mv ? oy * p.stride - 1 : -(1 << 20);
        a_ix[i] = ox * p.stride - 1;
      }
    }
  }
  unsigned b_off[BP];
#pragma unroll
  for (int i = 0; i < BP; ++i) {
    const int n = n0 + r0 + RPP * i;
    b_off[i] = (n < p.N) ? (unsigned)(((long)n * p.K + c8) * 2) : OOB;
  }

  typedef __attribute__((address_space(3))) void* lds_ptr;
  constexpr int LPT = AP + BP;                  // DMA instructions per wave per K tile
  const int nt = (p.dbg & 4) ? 0 : kt1 - kt0;

  // Request tile T (relative to kt0) into ring slot ST (compile-time): every wave writes 8 rows x 128 B (1 KiB,
  // lane-linear) per instruction.  Tiles past the end are requested out of bounds (zeros land in a dead slot), so
  // the DMA count per iteration is constant.
  int d_tap = 0, d_cc = 0;       // conv: (tap, channel offset) of the NEXT tile to request, advanced incrementally
  if (MODE != 0) {
    const int k0 = kt0 * BK;
    d_tap = min(k0 / p.Cin, 9);             // tap 9 = the folded skip connection's channel range
    d_cc = k0 - d_tap * p.Cin;
  }
  char* const dmaA = smem + wid * (8 * BK * 2);
  char* const dmaB = smem + NST * BM * BK * 2 + wid * (8 * BK * 2);
#define DF_DMA(T, ST)                                                                             \
  {                                                                                             \
    const bool live = (T) < nt;                                                                 \
    const unsigned k0b = (p.dbg & 1) ? 0u : (unsigned)(kt0 + (T)) * (BK * 2);                         \
    if (MODE == 0) {                                                                            \
      if (k0b < k1b) {                                                                          \
        const unsigned kb = live ? k0b : OOB;                                                   \
        _Pragma("unroll") for (int i = 0; i < AP; ++i)                                          \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(dmaA + ((ST) * BM + i * RPP) * (BK * 2)), 16, \
                                                   a_off[i] + kb, 0, 0, 0);                     \
      } else {   /* second tensor (wave-uniform branch) */                                      \
        const unsigned kb = k0b - k1b;                                                          \
        _Pragma("unroll") for (int i = 0; i < AP; ++i)                                          \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, (lds_ptr)(dmaA + ((ST) * BM + i * RPP) * (BK * 2)), 16, \
                                                   (live && a_off2[i] != OOB) ? a_off2[i] + kb : OOB, 0, 0, 0); \
      }                                                                                         \
    } else if (MODE == 3) {                                                                     \
      const unsigned delta = (unsigned)((((d_tap >> 1) * p.Wd + (d_tap & 1)) * p.lda + d_cc) * 2); \
      const unsigned bit = live ? (1u << d_tap) : 0u;                                           \
      _Pragma("unroll") for (int i = 0; i < AP; ++i)                                            \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(dmaA + ((ST) * BM + i * RPP) * (BK * 2)), 16, \
                                                 (a_msk[i] & bit) ? a_off[i] + delta : OOB, 0, 0, 0); \
    } else if (MODE == 1) {                                                                     \
      if (d_tap < 9) {                                                                          \
        const int ky = (d_tap * 11) >> 5, kx = d_tap - ky * 3;                                  \
        const unsigned delta = (unsigned)((((ky - 1) * p.Wd + (kx - 1)) * p.lda + d_cc) * 2);   \
        const unsigned bit = live ? (1u << d_tap) : 0u;                                         \
        _Pragma("unroll") for (int i = 0; i < AP; ++i)                                          \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(dmaA + ((ST) * BM + i * RPP) * (BK * 2)), 16, \
                                                   (a_msk[i] & bit) ? a_off[i] + delta : OOB, 0, 0, 0); \
      } else {   /* folded skip: centre pixel of the second tensor (wave-uniform branch) */      \
        const unsigned delta = (unsigned)(d_cc * 2);                                            \
        _Pragma("unroll") for (int i = 0; i < AP; ++i)                                          \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, (lds_ptr)(dmaA + ((ST) * BM + i * RPP) * (BK * 2)), 16, \
                                                   (live && a_off2[i] != OOB) ? a_off2[i] + delta : OOB, 0, 0, 0); \
      }                                                                                         \
    } else {                                                                                    \
      const int ky = (d_tap * 11) >> 5, kx = d_tap - ky * 3;                                    \
      _Pragma("unroll") for (int i = 0; i < AP; ++i) {                                          \
        const int uy = a_iy[i] + ky, ux = a_ix[i] + kx;                                         \
        const bool v = live && ((unsigned)uy < (unsigned)UH) && ((unsigned)ux < (unsigned)UW) && \
                       !(p.zstuff && ((uy | ux) & 1));                                          \
        const int sy = uy >> p.ups, sx = ux >> p.ups;                                           \
        const unsigned off = ((a_off[i] + (unsigned)(sy * p.Wd + sx)) * (unsigned)p.lda + (unsigned)(d_cc + c8)) * 2u; \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(dmaA + ((ST) * BM + i * RPP) * (BK * 2)), 16, \
                                                 v ? off : OOB, 0, 0, 0);                       \
      }                                                                                         \
    }                                                                                           \
    if (MODE != 0) {                                                                            \
      d_cc += BK;                                                                               \
      if (d_cc == p.Cin && d_tap < 9) {                                                         \
        d_cc = 0;                                                                               \
        ++d_tap;                                                                                \
      }                                                                                         \
    }                                                                                           \
    if constexpr (WD == 0) {                                                                    \
      const unsigned kb = live ? k0b : OOB;                                                     \
      _Pragma("unroll") for (int i = 0; i < BP; ++i)                                            \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(dmaB + ((ST) * BN + i * RPP) * (BK * 2)), 16, \
                                                 b_off[i] + kb, 0, 0, 0);                       \
    } else {   /* deep weight ring: tile T + WD into the running write slot */                  \
      DF_DMA_W((T) + WD);                                                                       \
    }                                                                                           \
  }
  // weight tile TW into the W ring's running write slot (deep-ring tiles only)
  int w_wr = 0, w_rd = 0;        // byte offsets of the W ring's write / read slot
#define DF_DMA_W(TW)                                                                              \
  {                                                                                             \
    const unsigned kbw = ((TW) < nt) ? ((p.dbg & 1) ? 0u : (unsigned)(kt0 + (TW)) * (BK * 2)) : OOB; \
    _Pragma("unroll") for (int i = 0; i < BP; ++i)                                              \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(dmaB + w_wr + i * RPP * (BK * 2)), 16, \
                                               b_off[i] + kbw, 0, 0, 0);                        \
    w_wr += WSLOT;                                                                              \
    if (w_wr == NSTB * WSLOT) w_wr = 0;                                                         \
  }

  // ---- prologue: fill NST-1 ring slots -- as early as the request offsets exist.  Everything below (accumulator init,
  // fragment offsets, epilogue prefetch: ~150-700 instructions) runs while the first tiles are in flight instead of in front of
  // them; the epilogue prefetch's loads are then younger than these requests, which only makes the first ring wait stricter.
  if constexpr (WD > 0) {     // the weight stream leads: tiles 0 .. WD-1 first, then the pairs [A(i), W(i + WD)] of the steady state
#pragma unroll
    for (int tw = 0; tw < WD; ++tw) DF_DMA_W(tw);
  }
  if (!PS || producer) {
    DF_DMA(0, 0);
    if (NST > 2) DF_DMA(1, 1);
    if (NST > 3) DF_DMA(2, 2);
    if (NST > 4) DF_DMA(3, 3);
  }
  __builtin_amdgcn_sched_barrier(0);
  stamp();

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- LDS fragment byte offsets (slot 0) for the 4 k-steps of a tile: loop invariant
  const char* fA[TM][4];
  const char* fB[TN][4];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = wm * WTM + i * 32 + l31;
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) fA[i][s2] = smem + (row * BK + (((2 * s2 + lh) ^ ((row >> 1) & 7)) << 3)) * 2;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * WTN + j * 32 + l31;
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2)
      fB[j][s2] = smem + NST * BM * BK * 2 + (row * BK + (((2 * s2 + lh) ^ ((row >> 1) & 7)) << 3)) * 2;
  }
#define DF_FRAG(DSTA, DSTB, S, ST)                                                                \
  {                                                                                             \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                              \
      DSTA[i] = *reinterpret_cast<const bf16x8*>(fA[i][S] + (ST) * BM * BK * 2);                \
    _Pragma("unroll") for (int j = 0; j < TN; ++j)                                              \
      DSTB[j] = *reinterpret_cast<const bf16x8*>(fB[j][S] + (WD == 0 ? (ST) * BN * BK * 2 : w_rd)); \
  }
#define DF_MMA(SRCA, SRCB, ACC)                                                                   \
  {                                                                                             \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                              \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                            \
        ACC[i][j] = DF_MFMA_32x32x16(SRCA[i], SRCB[j], ACC[i][j]); \
  }
  // One K tile out of ring slot ST; refills the slot of the previous tile ((ST + NST - 1) % NST).
  // The K-step boundary (counted vmcnt wait + s_barrier for tile it+1) sits BEFORE the last MFMA group of tile `it`, behind an
  // explicit s_waitcnt lgkmcnt(0): a ring slot may only be handed back to the DMA engine once every wave's ds_reads of it have
  // COMPLETED, not merely issued.  An LDS-DMA write is not ordered behind another wave's queued ds_read, and s_barrier does
  // not wait for LDS reads in flight; without the lgkmcnt wait the compiler sank the tail MFMAs (and the waits of their
  // operand reads) below the barrier, and a refill landing before a delayed read gave run-to-run differences when a second
  // process shared the CUs (tools/chk_probe.py: first diverging op always a halo conv; DESIGN.md section 4 "Determinism").
  // The last fragment reads are issued one MFMA group before the wait, so it normally costs nothing.
#define DF_RING_SYNC(VM)                                                                          \
  {                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* my reads of the slot about to be refilled are done */ \
    if constexpr (PS == 0) wait_vmcnt<DF_HUNT_VM(VM)>();   /* next tile landed: <= NST-2 younger tiles of this wave in flight */ \
    __builtin_amdgcn_s_barrier();    /* all parts of it visible; everyone is done with the current slot */  \
  }
  // (A three-fragment-set form with pinned issue order was measured slower for this kernel: 221 vs 224.5 steps/s.)
#define DF_ITER(ST)                                                                               \
  {                                                                                             \
    bf16x8 a0[TM], b0[TN], a1[TM], b1[TN];                                                      \
    DF_FRAG(a0, b0, 0, ST);                                                                     \
    if constexpr (PS == 0) DF_DMA(it + NST - 1, ((ST) + NST - 1) % NST);                        \
    DF_FRAG(a1, b1, 1, ST);                                                                     \
    DF_MMA(a0, b0, acc);                                                                        \
    DF_FRAG(a0, b0, 2, ST);                                                                     \
    DF_MMA(a1, b1, acc);                                                                        \
    DF_FRAG(a1, b1, 3, ST);                                                                     \
    DF_MMA(a0, b0, acc);                                                                        \
    DF_RING_SYNC((NST - 2) * LPT + (WD > 0 ? BP : 0));   /* deep W ring: the W request paired with A(it+1) is younger */ \
    DF_MMA(a1, b1, acc);                                                                        \
    ++it;                                                                                       \
    if constexpr (WD > 0) {                                                                     \
      w_rd += WSLOT;                                                                            \
      if (w_rd == NSTB * WSLOT) w_rd = 0;                                                       \
    }                                                                                           \
  }
  // epilogue roles: in a producer-specialised block BOTH wave sets run the store loops (2 * NT threads, block-linear index);
  // only the consumers park accumulators
  constexpr int NTE = PS ? 2 * NT : NT;
  const int etid = (int)threadIdx.x;
  const EpiVec ev = epi_prefetch<BN, NTE, EPI>(p, n0, etid);

  float2 ln_mr = make_float2(0.f, 1.f);
  if (ln_on && !producer) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LNS; ++i)
      if (i < p.ln_slots) {
        s1 += lnv[i].x;
        s2 += lnv[i].y;
      }
    const float inv = 1.0f / (float)p.ln_C;
    const float mean = s1 * inv;
    ln_mr = make_float2(mean, rsqrtf(fmaxf(s2 * inv - mean * mean, 0.f) + p.ln_eps));
  }

  if constexpr (PS != 0) {
    if (producer) {
      // the producer's K loop: request tile it + NST - 1 into the slot tile it - 1 was read from, wait until tile it + 1 has
      // landed (<= NST - 2 younger tiles of this wave in flight), meet the consumers at the K-step barrier
      wait_vmcnt<(NST - 2) * LPT>();
      __builtin_amdgcn_s_barrier();
#define DF_PITER(ST)                                                                              \
  {                                                                                             \
    DF_DMA(pit + NST - 1, ((ST) + NST - 1) % NST);                                              \
    wait_vmcnt<(NST - 2) * LPT>();                                                              \
    __builtin_amdgcn_s_barrier();                                                               \
    ++pit;                                                                                      \
  }
      int pit = 0;
      while (pit < nt) {
        DF_PITER(0);
        if (pit >= nt) break;
        DF_PITER(1);
        if (NST > 2) {
          if (pit >= nt) break;
          DF_PITER(2 % NST);
        }
        if (NST > 3) {
          if (pit >= nt) break;
          DF_PITER(3 % NST);
        }
        if (NST > 4) {
          if (pit >= nt) break;
          DF_PITER(4 % NST);
        }
      }
#undef DF_PITER
      wait_vmcnt<0>();       // the dead-slot requests of the last iterations land before the epilogue's first barrier
      if (p.dbg & 2) return;
      if (PS > 1 && wid >= WGM * WGN) {        // surplus producers: the two barriers of epilogue_block, nothing else
        epilogue_block_idle();
        return;
      }
      if constexpr (MODE != 3) {     // the store loops of the block epilogue (no accumulators to park)
        f32x16 none[TM][TN];
        bool vec_ok = true;
        if constexpr (EPI == EPI_ANY)
          vec_ok = !p.store_nchw && (p.N & 3) == 0 && (p.ldc & 3) == 0 && (p.ldr & 3) == 0 && (p.ld_rowbias & 3) == 0 &&
                   (p.res_bs & 3) == 0 && (p.c_bs & 3) == 0 && (p.ld_aux & 3) == 0;
        if (vec_ok)
          epilogue_block<BM, BN, NTE, TM, TN, EPI, false>(p, z, batch, reinterpret_cast<float*>(smem), none, 0, 0, n0, etid,
                                                          [&](int r) { return m0 + r; }, ev);
      }
      return;
    }
  }
  int it = 0;
  if constexpr (PS != 0) {
    // consumer K loop of a producer-specialised block: ONE loop body with a running ring-slot offset (the NST-times unrolled form
    // with its mid-body exits made the register allocator rotate the accumulator set between the unrolled bodies: copies and
    // spills inside the MFMA stream).  VALU is idle in this role, the v_add per fragment address is free.
    int sa = 0, sb = 0;                     // byte offsets of the current slot in the A / W rings
#define DF_FRAGD(DSTA, DSTB, S)                                                                   \
  {                                                                                             \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) DSTA[i] = *reinterpret_cast<const bf16x8*>(fA[i][S] + sa); \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) DSTB[j] = *reinterpret_cast<const bf16x8*>(fB[j][S] + sb); \
  }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();           // tile 0 landed (the producers waited for it) and visible
    stamp();
#pragma unroll 1
    for (; it < nt; ++it) {
      bf16x8 a0[TM], b0[TN], a1[TM], b1[TN];
      DF_FRAGD(a0, b0, 0);
      DF_FRAGD(a1, b1, 1);
      DF_MMA(a0, b0, acc);
      DF_FRAGD(a0, b0, 2);
      DF_MMA(a1, b1, acc);
      DF_FRAGD(a1, b1, 3);
      DF_MMA(a0, b0, acc);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads of this slot are done: it may be refilled
      __builtin_amdgcn_s_barrier();
      DF_MMA(a1, b1, acc);
      sa += BM * BK * 2;
      if (sa == NST * BM * BK * 2) sa = 0;
      sb += BN * BK * 2;
      if (sb == NST * BN * BK * 2) sb = 0;
    }
#undef DF_FRAGD
  } else {
  DF_RING_SYNC((NST - 2) * LPT + (WD > 0 ? BP : 0));       // tile 0 landed and visible
  stamp();
  while (it < nt) {
    DF_ITER(0);
    if (it >= nt) break;
    DF_ITER(1);
    if (NST > 2) {
      if (it >= nt) break;
      DF_ITER(2 % NST);
    }
    if (NST > 3) {
      if (it >= nt) break;
      DF_ITER(3 % NST);
    }
    if (NST > 4) {
      if (it >= nt) break;
      DF_ITER(4 % NST);
    }
  }
  }
  if constexpr (PS == 0) wait_vmcnt<0>();   // dead-slot requests of the last iterations must land before LDS is released
  stamp();

  // ---- epilogue
  if (p.dbg & 2) {   // tools: keep the accumulators alive, store nothing
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  if constexpr (MODE == 3) {
    // tile row (input-resolution pixel n, y, x) -> output pixel (n, 2y + a, 2x + b) of the x2 map; the epilogue sees the
    // OUTPUT tensor: M = 4 * rows.  Row table behind the epilogue tile (launch_cfg reserves BM ints).
    int* srow = reinterpret_cast<int*>(smem + (size_t)BM * (BN + 4) * 4 + (size_t)BM * 8);
    __builtin_amdgcn_s_barrier();
    for (int rr = etid; rr < BM; rr += NTE) {
      const int m = m0 + rr, hw = p.OH * p.OW;
      const int nb = m / hw, rem = m - nb * hw, oy = rem / p.OW, ox = rem - oy * p.OW;
      srow[rr] = (m < p.M) ? ((nb * 2 * p.OH + 2 * oy + (phase >> 1)) * 2 * p.OW + 2 * ox + (phase & 1)) : 4 * p.M;
    }
    __syncthreads();
    GemmParams pe = p;
    pe.M = 4 * p.M;
    auto rowmap = [&](int rr) { return srow[rr]; };
    if constexpr (EPI != EPI_ANY) {
      epilogue_block<BM, BN, NTE, TM, TN, EPI>(pe, z, 0, reinterpret_cast<float*>(smem), acc, wm * WTM, wn * WTN, n0, etid, rowmap, ev);
    } else {
      const bool vec_ok = !p.store_nchw && (p.N & 3) == 0 && (p.ldc & 3) == 0 && (p.ldr & 3) == 0 && (p.ld_rowbias & 3) == 0 &&
                          (p.ld_aux & 3) == 0;
      if (vec_ok) {
        epilogue_block<BM, BN, NTE, TM, TN, EPI>(pe, z, 0, reinterpret_cast<float*>(smem), acc, wm * WTM, wn * WTN, n0, etid, rowmap, ev);
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          int rowv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) rowv[r] = rowmap(wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
          epilogue_band<TN>(pe, z, 0, rowv, acc[i], n0 + wn * WTN, l31);
        }
      }
    }
  } else if constexpr (EPI != EPI_ANY) {     // the host routes only vectorisable, row-major problems to the specialised kernels
    epilogue_block<BM, BN, NTE, TM, TN, EPI>(p, z, batch, reinterpret_cast<float*>(smem), acc, wm * WTM, wn * WTN, n0, etid,
                                             [&](int r) { return m0 + r; }, ev, ln_mr);
  } else {
    const bool vec_ok = !p.store_nchw && (p.N & 3) == 0 && (p.ldc & 3) == 0 && (p.ldr & 3) == 0 &&
                        (p.ld_rowbias & 3) == 0 && (p.res_bs & 3) == 0 && (p.c_bs & 3) == 0 && (p.ld_aux & 3) == 0;
    if (vec_ok) {
      epilogue_block<BM, BN, NTE, TM, TN, EPI>(p, z, batch, reinterpret_cast<float*>(smem), acc, wm * WTM, wn * WTN, n0,
                                               etid, [&](int r) { return m0 + r; }, ev, ln_mr);
      return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int rowv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) rowv[r] = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      epilogue_band<TN>(p, z, batch, rowv, acc[i], n0 + wn * WTN, l31);
    }
  }
#endif
}


// Runtime-valued vmcnt wait (the immediate must be a literal): small switch over the values that occur.
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
  switch (n) {
    case 1: wait_vmcnt<1>(); break;
    case 2: wait_vmcnt<2>(); break;
    case 3: wait_vmcnt<3>(); break;
    case 4: wait_vmcnt<4>(); break;
    case 5: wait_vmcnt<5>(); break;
    case 6: wait_vmcnt<6>(); break;
    case 7: wait_vmcnt<7>(); break;
    case 8: wait_vmcnt<8>(); break;
    case 9: wait_vmcnt<9>(); break;
    case 10: wait_vmcnt<10>(); break;
    case 11: wait_vmcnt<11>(); break;
    case 12: wait_vmcnt<12>(); break;
    case 13: wait_vmcnt<13>(); break;
    case 14: wait_vmcnt<14>(); break;
    case 15: wait_vmcnt<15>(); break;
    case 16: wait_vmcnt<16>(); break;
    case 17: wait_vmcnt<17>(); break;
    case 18: wait_vmcnt<18>(); break;
    case 19: wait_vmcnt<19>(); break;
    case 20: wait_vmcnt<20>(); break;
    case 21: wait_vmcnt<21>(); break;
    case 22: wait_vmcnt<22>(); break;
    case 23: wait_vmcnt<23>(); break;
    case 24: wait_vmcnt<24>(); break;
    case 25: wait_vmcnt<25>(); break;
    case 26: wait_vmcnt<26>(); break;
    case 27: wait_vmcnt<27>(); break;
    case 28: wait_vmcnt<28>(); break;
    case 29: wait_vmcnt<29>(); break;
    case 30: wait_vmcnt<30>(); break;
    case 31: wait_vmcnt<31>(); break;
    case 32: wait_vmcnt<32>(); break;
    case 33: wait_vmcnt<33>(); break;
    case 34: wait_vmcnt<34>(); break;
    case 35: wait_vmcnt<35>(); break;
    case 36: wait_vmcnt<36>(); break;
    case 37: wait_vmcnt<37>(); break;
    case 38: wait_vmcnt<38>(); break;
    case 39: wait_vmcnt<39>(); break;
    case 40: wait_vmcnt<40>(); break;
    default: wait_vmcnt<0>(); break;   // conservative: full drain
  }
}

// ===============================================================================================================
// conv3x3 (stride 1, pad 1) with an LDS-staged HALO tile.
//
// The implicit-GEMM kernel above re-fetches the A operand once per tap (9x).  On this chip the L2->LDS fill rate of
// a CU (~20 B/clk measured) is what bounds these GEMMs, so here a block owns PB spatial patches of TH x TW output
// pixels (BM = PB*TH*TW), stages their (TH+2) x (TW+2) input halo for a 64-channel slice ONCE, and runs all 9 taps
// out of it: per slice the block fetches HR*128 B of activations + 9 * BN*128 B of weights for 9*2*BM*BN*64 FLOP.
// Pipeline: A halo double-buffered (slice c+1 requested during slice c), weights in a 4-deep ring requested 3 taps
// ahead; the 9 taps are unrolled so every wait is a compile-time `vmcnt` (loads retire in issue order):
//   wait for W(c,t):  younger = W(+1) .. W(+NSTW-2) and, for t in 1..NSTW-1, the A(c+1) request issued at tap 0.
// Zero padding and ragged edges come from out-of-bounds buffer offsets (hardware writes zeros to LDS).
// PS > 0: producer-specialised block (see gemm_bf16_kernel): PS producer wavefronts per consumer wavefront issue every halo and
// weight request in the symmetric kernel's order (so the counted waits are the same); consumers read fragments and run MFMAs.
template <int BM, int BN, int WGM, int WGN, int NSTW, int EPI, int PS>
__device__ __forceinline__ void conv3x3_halo_body(const GemmParams& p);
template <int BM, int BN, int WGM, int WGN, int NSTW, int EPI, int PS = 0>
__global__ __launch_bounds__(64 * WGM * WGN * (1 + PS)) void conv3x3_halo_kernel(GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  const DfTouch ka = gemm_kernarg_touch();
  conv3x3_halo_body<BM, BN, WGM, WGN, NSTW, EPI, PS>(p);
  gemm_kernarg_touch_end(ka);
#endif
}
template <int BM, int BN, int WGM, int WGN, int NSTW, int EPI, int PS>
__device__ __forceinline__ void conv3x3_halo_body(const GemmParams& p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NT = 64 * WGM * WGN;          // threads (PS: the consumer threads)
  constexpr int NTP = PS ? NT * PS : NT;      // threads that issue the DMA requests
  constexpr int RPP = NTP / 8;                // LDS rows written per DMA pass of the whole block
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int WPASS = (BN + RPP - 1) / RPP; // weight passes per tap
  constexpr int A2P = BM / RPP;               // passes of one folded-skip tail slice (BM centre pixels)
  static_assert(NSTW >= 3 && NSTW <= 8, "weight ring depth");
  static_assert(PS == 0 || (BM % RPP == 0 && A2P <= 8), "tail slice passes");
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;

  const int raw_wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const bool producer = PS && raw_wid >= WGM * WGN;            // wave-uniform role; tid / wid are role-local
  const int tid = producer ? (int)threadIdx.x - NT : (int)threadIdx.x, lane = tid & 63, wid = producer ? raw_wid - WGM * WGN : raw_wid;
  const int wm = wid / WGN, wn = wid % WGN;
  const int l31 = lane & 31, lh = lane >> 5;
  constexpr int NTE = PS ? 2 * NT : NT;       // threads of the epilogue's store loops (consumers + the first WGM x WGN producers)
  const int etid = (int)threadIdx.x;

  // tools (p.dbg bit 6, split-K 1): thread 0 stamps the shader clock at phase boundaries into p.partial[block][32] (uint64)
  int n_stamp = 0;
  auto stamp = [&]() {
    if ((p.dbg & 64) && etid == 0 && n_stamp < 32)
      reinterpret_cast<unsigned long long*>(p.partial)[(long)blockIdx.x * 32 + n_stamp++] = __builtin_amdgcn_s_memtime();
  };
  stamp();
  const int TH = p.th, TW = p.tw, HWp = (TH + 2) * (TW + 2), PPX = TH * TW;
  const int PB = BM / PPX;                      // patches per block
  const int HR = PB * HWp;                      // halo rows
  const int APASS = (HR + RPP - 1) / RPP;       // halo passes (block-uniform)
  const int HRP = APASS * RPP;
  bf16_t* sA = reinterpret_cast<bf16_t*>(smem);                 // [2][HRP][64]
  bf16_t* sW = sA + 2 * HRP * BK;                               // [NSTW][max(BN,RPP)][64]
  constexpr int WROWS = WPASS * RPP;

  const int npx = p.Wd / TW, npy = p.H / TH;    // patches per image
  const int npatch = (p.M / PPX);               // total patches (M = NB*H*W)
  const int nbm = (npatch + PB - 1) / PB, nbn = (p.N + BN - 1) / BN;
  const int nblk = nbm * nbn;
  int lid;
  {
    const int bid = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int mt, nt_;
  tile_of(lid, nbm, nbn, p.gm, mt, nt_);
  const int n0 = nt_ * BN;

  const int z = blockIdx.z;
  const int nchunk = p.Cin / BK;
  int c0 = 0, c1 = nchunk;
  if (p.splitk > 1) {
    const int per = (nchunk + p.splitk - 1) / p.splitk;
    c0 = z * per;
    c1 = min(nchunk, c0 + per);
  }
  const int nc = max(c1 - c0, 0);
  // ResBlock skip connection folded into conv2 (openai_unetmodel.py:234-241, 275: skip(x) + h): a TAIL of n2 one-tap K steps after
  // the nine-tap slices, reading 64-channel slices [s0, s0 + n2) of the second tensor at the centre pixel (producer-specialised
  // tiles only, round 5).  With split-K the skip slices are dealt to the splits that own conv slices, in the same order.
  int s0 = 0, n2 = 0;
  if (PS != 0 && p.Cin2 > 0) {
    const int n2tot = p.Cin2 / BK;
    if (p.splitk > 1) {
      const int per = (nchunk + p.splitk - 1) / p.splitk, used = (nchunk + per - 1) / per;
      const int per2 = (n2tot + used - 1) / used;
      s0 = z * per2;
      n2 = (z < used) ? max(min(n2tot, s0 + per2) - s0, 0) : 0;
    } else {
      n2 = n2tot;
    }
  }
  const int nc9 = nc * 9;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(PS != 0 && p.A2 ? p.A2 : p.A), 0,
                                                                          (int)(PS != 0 && p.A2 ? p.a2_bytes : 0u), 0x00020000);

  // ---- staging coordinates.  Thread (r0 = tid>>3, slot = tid&7) fills LDS slot `slot` of rows r0 + RPP*i with the
  // source chunk slot ^ ((row>>1)&7)   (RPP is a multiple of 16, so the swizzle term is the same for every pass).
  const int r0 = tid >> 3;
  const int c8 = ((tid & 7) ^ ((r0 >> 1) & 7)) * 8;      // weight rows: consecutive rows, swizzle by row pair
  // Halo rows are read by the MFMA fragments in runs of TW pixels of consecutive image rows, so a swizzle by the LINEAR halo
  // row ((hr >> 1) & 7) puts two of the 16 lanes of a ds_read_b128 group on one 16-B slot (2-way conflicts on ~40 % of the
  // LDS cycles, SQ_LDS_BANK_CONFLICT).  The halo slot is swizzled by the pixel's x-pair plus SC * (image row counter):
  // conflict-free for TW = 16 (SC = 0) and TW = 8 (SC = 4) at every tap shift.
  const int SC = (TW == 16) ? 0 : 4;
  // all quotients below are of values < 2^22 (launch_halo checks the patch count): FastDiv instead of 32-bit division
  const FastDiv fd_hwp(HWp), fd_tw2(TW + 2), fd_pimg(npy * npx), fd_npx(npx), fd_ppx(PPX), fd_tw(TW);
  constexpr int MAXAP = 12;
  // Halo row -> source pixel, ONE decomposition per halo row of the block (round 4): every thread of the block (both roles)
  // resolves the rows hr = thread, thread + block, ... into a table entry (byte offset of the pixel's channel 0 | 3-bit swizzle
  // key, or ~0 outside the image) in LDS behind the ring; the requesting threads then pick up their <= 12 rows with one ds_read
  // and 4 VALU instructions each.  (Before: 12 x (3 FastDiv chains + bounds) = ~540 VALU instructions per thread in front of the
  // first request -- tools/halo_stamps.py: 5.4 k cycles from kernel start to the end of the offset arithmetic.)
  unsigned* atab = reinterpret_cast<unsigned*>(smem + p.halo_ring_bytes);
  for (int hr = (int)threadIdx.x; hr < HR; hr += NT * (1 + PS)) {
    const int pi = fd_hwp.div(hr), rem = hr - pi * HWp;
    const int hy = fd_tw2.div(rem), hx = rem - hy * (TW + 2);
    const int g = mt * PB + pi;                  // global patch id
    unsigned e = 0xffffffffu;
    if (g < npatch) {
      const int n = fd_pimg.div(g), gr = g - n * (npy * npx);
      const int gy = fd_npx.div(gr), gx = gr - gy * npx;
      const int y = gy * TH + hy - 1, x = gx * TW + hx - 1;
      if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.Wd)
        e = (unsigned)((((long)(n * p.H + y) * p.Wd + x) * p.lda) * 2) | (unsigned)(((hx >> 1) + SC * (pi * (TH + 2) + hy)) & 7);
    }
    atab[hr] = e;
  }
  // (not __syncthreads(): its release fence waits for every outstanding vector-memory operation the compiler knows of -- since
  // round 6 that includes the kernel-entry touch loads, i.e. a cold code line in front of the first operand request.  The table
  // is LDS: the writers' lgkmcnt(0) + the barrier order it for every reader.)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  unsigned a_off[MAXAP];
#pragma unroll
  for (int i = 0; i < MAXAP; ++i) {
    const int hr = r0 + RPP * i;                 // halo row
    unsigned off = OOB;
    if (i < APASS && hr < HR) {      // (PS: the consumers take the odd passes of the FIRST halo, see the prologue)
      const unsigned e = atab[hr];
      if (e != 0xffffffffu) off = (e & ~7u) + ((((unsigned)tid & 7u) ^ (e & 7u)) << 4);     // lda % 8 == 0: the low 4 bits of a pixel offset are free
    }
    a_off[i] = off;
  }
  unsigned w_off[WPASS];
#pragma unroll
  for (int i = 0; i < WPASS; ++i) {
    const int n = n0 + r0 + RPP * i;
    w_off[i] = (n < p.N && r0 + RPP * i < BN) ? (unsigned)(((long)n * p.K + c8) * 2) : OOB;
  }

  // DMA helpers -------------------------------------------------------------------------------------------------
  // A halo of channel slice C (absolute slice index) into buffer BUF
#define DF_HALO_A(C, BUF)                                                                         \
  {                                                                                             \
    bf16_t* a_ = sA + (BUF) * HRP * BK + wid * (8 * BK);                                        \
    const bool live = (C) < c1;                                                                 \
    const unsigned cb = (unsigned)(C) * (BK * 2);                                               \
    _Pragma("unroll") for (int i = 0; i < MAXAP; ++i)                                           \
      if (i < APASS)                                                                            \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(a_ + i * RPP * BK), 16,         \
                                                 (live && a_off[i] != OOB) ? a_off[i] + cb : OOB, 0, 0, 0); \
  }
  // weights of iteration IT (= slice*9 + tap, relative to c0) into ring slot ST
#define DF_HALO_W(IT, ST)                                                                         \
  {                                                                                             \
    bf16_t* w_ = sW + (ST) * WROWS * BK + wid * (8 * BK);                                       \
    const int cs_ = (IT) / 9, tp_ = (IT) - cs_ * 9;                                             \
    const bool live = (IT) < nc9 + n2;          /* (IT) >= nc9: weight tile of step (IT) - nc9 of the folded-skip tail */ \
    const unsigned kb = (IT) < nc9 ? (unsigned)(tp_ * p.Cin + (c0 + cs_) * BK) * 2u             \
                                   : (unsigned)(9 * p.Cin + (s0 + (IT) - nc9) * BK) * 2u;      \
    _Pragma("unroll") for (int i = 0; i < WPASS; ++i)                                           \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(w_ + i * RPP * BK), 16,           \
                                               (live && w_off[i] != OOB) ? w_off[i] + kb : OOB, 0, 0, 0); \
  }

  // ---- prologue: A(0), W(0 .. NSTW-2) -- requested as soon as the offsets exist; fragment coordinates, accumulator init and
  // the epilogue prefetch below run while these are in flight
  // PS: the first halo is the longest request burst of the block (<= 12 per wavefront in front of the first MFMA) and the consumers
  // have nothing to do yet: they take its odd passes (own vmcnt(0) in front of the first tap barrier), the producers the even ones.
#define DF_HALO_A_PAR(C, BUF, PAR)                                                                \
  {                                                                                             \
    bf16_t* a_ = sA + (BUF) * HRP * BK + wid * (8 * BK);                                        \
    const unsigned cb = (unsigned)(C) * (BK * 2);                                               \
    _Pragma("unroll") for (int i = (PAR); i < MAXAP; i += 2)                                    \
      if (i < APASS)                                                                            \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(a_ + i * RPP * BK), 16,         \
                                                 ((C) < c1 && a_off[i] != OOB) ? a_off[i] + cb : OOB, 0, 0, 0); \
  }
  if constexpr (PS != 0) {
    if (producer) {
      DF_HALO_A_PAR(c0, 0, 0);
#pragma unroll
      for (int t = 0; t < NSTW - 1; ++t) DF_HALO_W(t, t);
    } else {
      DF_HALO_A_PAR(c0, 0, 1);
    }
  } else {
    DF_HALO_A(c0, 0);
#pragma unroll
    for (int t = 0; t < NSTW - 1; ++t) DF_HALO_W(t, t);
  }
#undef DF_HALO_A_PAR
  __builtin_amdgcn_sched_barrier(0);
  stamp();

  // ---- MFMA row -> halo row of tap (0,0) and output pixel index
  int hb[TM], hbx[TM], hbq[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = wm * WTM + i * 32 + l31;
    const int pi = fd_ppx.div(r), rem = r - pi * PPX;
    const int y = fd_tw.div(rem), x = rem - y * TW;
    hb[i] = pi * HWp + y * (TW + 2) + x;
    hbx[i] = x;
    hbq[i] = SC * (pi * (TH + 2) + y);
  }
  int fb[TN], sb[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * WTN + j * 32 + l31;
    fb[j] = row * BK;
    sb[j] = (row >> 1) & 7;
  }

  const EpiVec ev = epi_prefetch<BN, NTE, EPI>(p, n0, etid);
  // tile row -> NHWC pixel index of its output pixel (epilogue).  Five integer divisions per row: computed once per row into
  // a BM-entry LDS table behind the operand ring instead of once per 4-column chunk of the epilogue loop.
  auto rowmap_calc = [&](int rr) {
    const int pi = fd_ppx.div(rr), rem = rr - pi * PPX;
    const int y = fd_tw.div(rem), x = rem - y * TW;
    const int g = mt * PB + pi;
    const int n = fd_pimg.div(g), gr = g - n * (npy * npx);
    const int gy = fd_npx.div(gr), gx = gr - gy * npx;
    return (g < npatch) ? (n * p.H + gy * TH + y) * p.Wd + gx * TW + x : p.M;
  };
  int* srow = reinterpret_cast<int*>(smem + p.halo_ring_bytes);
  auto rowmap = [&](int rr) { return srow[rr]; };

  if constexpr (PS != 0) {
    if (producer) {
      // producer K loop: per tap, the weight request NSTW - 1 taps ahead (+ at tap 0 the next slice's halo), the counted wait for
      // the next tap's operands -- DF_HALO_SYNC's rule, the issue order is the symmetric kernel's -- and the tap barrier
      wait_vmcnt<(NSTW - 2) * WPASS>();
      __builtin_amdgcn_s_barrier();
      // The next slice's halo is requested in pieces, pass i with tap i % 8, IN FRONT of that tap's weight request (round 4,
      // tools/halo_stamps.py: issued whole at tap 0, its <= 12 requests per wavefront at ~130 cycles each made tap 0 cost 1.7 k
      // cycles while taps 1-8 ran at ~480 -- the consumers sat at tap 0's barrier).  Loads retire in issue order, so with a(T) halo
      // requests at tap T the operands of tap T + 1 (weights requested NSTW - 2 = 2 taps earlier) have landed once at most
      // 2 * WPASS + a(T - 1) + a(T) younger requests are outstanding; the wait in front of the next slice's tap 0 allows the two
      // youngest weight tiles only, which covers the whole halo (a(8) = 0 and a(7) sits in front of tap 7's weight request).
      static_assert(NSTW == 4, "the spread halo schedule counts on weights requested 3 taps ahead");
      int acnt[10], acnt2[10];
#pragma unroll
      for (int t = 0; t < 10; ++t) {
        acnt[t] = (t >= 1 && t <= 8) ? (APASS > t - 1) + (APASS > t + 7) : 0;      // acnt[T + 1] = a(T)
        acnt2[t] = (t >= 1 && t <= 8) ? (A2P > t - 1) : 0;                         // the same for the first tail slice (A2P <= 8 passes)
      }
      // Folded-skip tail (round 5).  Tail step j multiplies the BM centre pixels of slice s0 + j of the second tensor (LDS rows =
      // tile rows, 128 B each, swizzled like the generic kernel's A tiles) with weight tile nc9 + j of the ring.  The tail's
      // activation slots are three BM-row windows of the two halo buffers (3 BM <= 2 HRP, checked by the launcher); slot 0 lies
      // inside the buffer that is FREE during the last nine-tap slice and is requested in that slice's spread schedule in place
      // of the (dead) next halo; slots 1 / 2 reach into the buffer the last slice reads and are requested by tail step 0, after
      // its barrier.  From then on step j requests slice j + 2 into the slot step j - 1 read; the wait in front of step j + 1
      // allows slice j + 2 and the two youngest weight tiles.
      unsigned a2_off[A2P];
#pragma unroll
      for (int i = 0; i < A2P; ++i) {
        const int px = rowmap_calc(r0 + RPP * i);
        a2_off[i] = (n2 > 0 && px < p.M) ? (unsigned)(((long)px * p.lda2 + c8) * 2) : OOB;
      }
      const int tfree = nc & 1;                // halo buffer that is free during the last nine-tap slice
      auto tslot = [&](int k) { return (tfree == 0 ? k * BM : 2 * HRP - (k + 1) * BM) * (BK * 2); };   // byte offset of tail slot k in sA
#define DF_TAIL_A(J, SLOT)                                                                        \
  {                                                                                             \
    char* a_ = reinterpret_cast<char*>(sA) + tslot(SLOT) + wid * (8 * BK * 2);                  \
    const bool live = (J) < n2;                                                                 \
    const unsigned cb = (unsigned)(s0 + (J)) * (BK * 2);                                        \
    _Pragma("unroll") for (int i = 0; i < A2P; ++i)                                             \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, (lds_ptr)(a_ + i * RPP * BK * 2), 16,      \
                                               (live && a2_off[i] != OOB) ? a2_off[i] + cb : OOB, 0, 0, 0); \
  }
#define DF_HALO_A_PART(C, BUF, T)                                                                 \
  {                                                                                             \
    bf16_t* a_ = sA + (BUF) * HRP * BK + wid * (8 * BK);                                        \
    const bool live = (C) < c1;                                                                 \
    const unsigned cb = (unsigned)(C) * (BK * 2);                                               \
    _Pragma("unroll") for (int i = (T); i < MAXAP; i += 8)                                      \
      if ((T) < 8 && i < APASS)                                                                 \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(a_ + i * RPP * BK), 16,         \
                                                 (live && a_off[i] != OOB) ? a_off[i] + cb : OOB, 0, 0, 0); \
  }
#define DF_PTAP(T)                                                                                \
  {                                                                                             \
    const int it_ = cs * 9 + (T);                                                               \
    if (tail0) {              /* pass T of the first tail slice into tail slot 0 */              \
      if ((T) < A2P) {                                                                          \
        char* a_ = reinterpret_cast<char*>(sA) + tslot(0) + wid * (8 * BK * 2);                 \
        const unsigned o_ = a2_off[(T) < A2P ? (T) : 0];                                        \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, (lds_ptr)(a_ + (T) * RPP * BK * 2), 16,  \
                                                 o_ != OOB ? o_ + (unsigned)s0 * (BK * 2) : OOB, 0, 0, 0); \
      }                                                                                         \
    } else {                                                                                    \
      DF_HALO_A_PART(c0 + cs + 1, (cs + 1) & 1, T);                                             \
    }                                                                                           \
    DF_HALO_W(it_ + NSTW - 1, (it_ + NSTW - 1) % NSTW);                                         \
    if ((T) == 8) wait_vmcnt<(NSTW - 2) * WPASS>();                                             \
    else wait_vmcnt_dyn((NSTW - 2) * WPASS + (tail0 ? acnt2[(T)] + acnt2[(T) + 1] : acnt[(T)] + acnt[(T) + 1])); \
    __builtin_amdgcn_s_barrier();                                                               \
  }
      for (int cs = 0; cs < nc; ++cs) {
        const bool tail0 = n2 > 0 && cs == nc - 1;
        DF_PTAP(0) DF_PTAP(1) DF_PTAP(2) DF_PTAP(3) DF_PTAP(4) DF_PTAP(5) DF_PTAP(6) DF_PTAP(7) DF_PTAP(8)
      }
      if (n2 > 0) {
        // tail step 0: slices 1 and 2 (their slots were under the last nine-tap slice until its last barrier) + weight tile nc9 + 3;
        // step 1 needs slice 1: slice 2 and that weight tile may still be in flight
        DF_TAIL_A(1, 1);
        DF_TAIL_A(2, 2);
        DF_HALO_W(nc9 + NSTW - 1, (nc9 + NSTW - 1) % NSTW);
        wait_vmcnt<A2P + WPASS>();
        __builtin_amdgcn_s_barrier();
        int ts = 0;                            // slot of slice j + 2 = (j + 2) % 3
        for (int j = 1; j < n2; ++j) {
          DF_TAIL_A(j + 2, ts);
          DF_HALO_W(nc9 + j + NSTW - 1, (nc9 + j + NSTW - 1) % NSTW);
          wait_vmcnt<(NSTW - 2) * WPASS + A2P>();
          __builtin_amdgcn_s_barrier();
          ts = (ts == 2) ? 0 : ts + 1;
        }
      }
#undef DF_TAIL_A
#undef DF_PTAP
#undef DF_HALO_A_PART
      wait_vmcnt<0>();
      if (PS > 1 && wid >= WGM * WGN) {      // surplus producers: the row-table barrier and epilogue_block's two, nothing else
        __builtin_amdgcn_s_barrier();
        epilogue_block_idle();
        return;
      }
      for (int rr = etid; rr < BM; rr += NTE) srow[rr] = rowmap_calc(rr);
      __syncthreads();
      bool vec_ok = true;
      if constexpr (EPI == EPI_ANY)
        vec_ok = !p.store_nchw && (p.N & 3) == 0 && (p.ldc & 3) == 0 && (p.ldr & 3) == 0 && (p.ld_rowbias & 3) == 0 && (p.ld_aux & 3) == 0;
      if (vec_ok) {
        f32x16 none[TM][TN];
        epilogue_block<BM, BN, NTE, TM, TN, EPI, false>(p, z, 0, reinterpret_cast<float*>(smem), none, 0, 0, n0, etid, rowmap, ev);
      }
      return;
    }
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Wait for the operands of tap T (W(c,T); at T = 0 also the halo of slice c) and hand the ring slot / halo buffer read by
  // the previous tap back to the DMA engine.  VM = loads of this wave allowed to be still in flight.  The lgkmcnt(0) makes
  // sure this wave's ds_reads of the slot to be refilled have COMPLETED (see DF_RING_SYNC of the generic kernel).
#define DF_HALO_SYNC(T)                                                                           \
  {                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                          \
    if constexpr (PS == 0) {                                                                    \
      if ((T) >= 1 && (T) <= NSTW - 1) wait_vmcnt_dyn((NSTW - 2) * WPASS + APASS);             \
      else wait_vmcnt<(NSTW - 2) * WPASS>();                                                    \
    }                                                                                           \
    __builtin_amdgcn_s_barrier();                                                               \
  }
#define DF_HALO_READ(DA, DB, S)                                                                   \
  {                                                                                             \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                              \
      DA[i] = *reinterpret_cast<const bf16x8*>(a + ha_[i] + (((2 * (S) + lh) ^ sa_[i]) << 3));  \
    _Pragma("unroll") for (int j = 0; j < TN; ++j)                                              \
      DB[j] = *reinterpret_cast<const bf16x8*>(b + fb[j] + (((2 * (S) + lh) ^ sb[j]) << 3));    \
  }
#define DF_HALO_MMA(SA, SB)                                                                       \
  {                                                                                             \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                              \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                            \
        acc[i][j] = DF_MFMA_32x32x16(SA[i], SB[j], acc[i][j]);                                  \
  }
  // One tap: the boundary to the NEXT tap sits before the last MFMA group (its fragment reads were issued one group earlier).
#define DF_TAP(T)                                                                        \
  {                                                                                             \
    const int it_ = cs * 9 + (T);                                                               \
    const bf16_t* a = sA + (cs & 1) * HRP * BK;                                                 \
    const bf16_t* b = sW + (it_ % NSTW) * WROWS * BK;                                           \
    constexpr int dy_ = (T) / 3, dx_ = (T) - dy_ * 3;                                           \
    int ha_[TM], sa_[TM];                                                                       \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                            \
      const int hr = hb[i] + dy_ * (TW + 2) + dx_;                                              \
      ha_[i] = hr * BK;                                                                         \
      sa_[i] = (((hbx[i] + dx_) >> 1) + hbq[i] + SC * dy_) & 7;                                 \
    }                                                                                           \
    bf16x8 af0[TM], bf0[TN], af1[TM], bf1[TN], af2[TM], bf2[TN];                                \
    DF_HALO_READ(af0, bf0, 0);                                                                  \
    DF_HALO_READ(af1, bf1, 1);                                                                  \
    DF_HALO_READ(af2, bf2, 2);                                                                  \
    DF_HALO_MMA(af0, bf0);                                                                      \
    DF_HALO_READ(af0, bf0, 3);                                                                  \
    __builtin_amdgcn_sched_barrier(0);   /* the last reads are ISSUED here: two MFMA groups of cover before the wait */ \
    if constexpr (PS == 0) {                                                                    \
      DF_HALO_W(it_ + NSTW - 1, (it_ + NSTW - 1) % NSTW);   /* refill requests interleave with the MFMA-only stretch */ \
      if ((T) == 0) DF_HALO_A(c0 + cs + 1, (cs + 1) & 1);                                       \
    }                                                                                           \
    DF_HALO_MMA(af1, bf1);                                                                      \
    DF_HALO_MMA(af2, bf2);                                                                      \
    __builtin_amdgcn_sched_barrier(0);   /* ... and those MFMA groups stay in front of it */           \
    DF_HALO_SYNC(((T) + 1) % 9);                                                                \
    DF_HALO_MMA(af0, bf0);                                                                      \
  }

  stamp();
  if constexpr (PS != 0) wait_vmcnt<0>();      // this consumer's share of the first halo
  DF_HALO_SYNC(0);
  stamp();
  for (int cs = 0; cs < nc; ++cs) {
    DF_TAP(0)
    DF_TAP(1)
    DF_TAP(2)
    DF_TAP(3)
    DF_TAP(4)
    DF_TAP(5)
    DF_TAP(6)
    DF_TAP(7)
    DF_TAP(8)
    stamp();
  }
  if constexpr (PS != 0) {
    // folded-skip tail (see the producer's side): one K step per 64-channel slice of the second tensor, centre pixels only
    if (n2 > 0) {
      const int tfree = nc & 1;
      const int sat = (l31 >> 1) & 7;
      int ts = 0;
      for (int j = 0; j < n2; ++j) {
        const char* a = reinterpret_cast<const char*>(sA) + (tfree == 0 ? ts * BM : 2 * HRP - (ts + 1) * BM) * (BK * 2) +
                        (wm * WTM + l31) * (BK * 2);
        const bf16_t* b = sW + ((nc9 + j) % NSTW) * WROWS * BK;
#define DF_TAIL_READ(DA, DB, S)                                                                   \
  {                                                                                             \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                              \
      DA[i] = *reinterpret_cast<const bf16x8*>(a + i * 32 * (BK * 2) + (((2 * (S) + lh) ^ sat) << 4)); \
    _Pragma("unroll") for (int jj = 0; jj < TN; ++jj)                                           \
      DB[jj] = *reinterpret_cast<const bf16x8*>(b + fb[jj] + (((2 * (S) + lh) ^ sb[jj]) << 3)); \
  }
        bf16x8 af0[TM], bf0[TN], af1[TM], bf1[TN], af2[TM], bf2[TN];
        DF_TAIL_READ(af0, bf0, 0);
        DF_TAIL_READ(af1, bf1, 1);
        DF_TAIL_READ(af2, bf2, 2);
        DF_HALO_MMA(af0, bf0);
        DF_TAIL_READ(af0, bf0, 3);
        __builtin_amdgcn_sched_barrier(0);
        DF_HALO_MMA(af1, bf1);
        DF_HALO_MMA(af2, bf2);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        DF_HALO_MMA(af0, bf0);
#undef DF_TAIL_READ
        ts = (ts == 2) ? 0 : ts + 1;
      }
      stamp();
    }
  }
  if constexpr (PS == 0) wait_vmcnt<0>();
  stamp();

  // ---- epilogue (row table: see rowmap_calc above)
  for (int rr = etid; rr < BM; rr += NTE) srow[rr] = rowmap_calc(rr);
  __syncthreads();
  if constexpr (EPI != EPI_ANY) {
    epilogue_block<BM, BN, NTE, TM, TN, EPI>(p, z, 0, reinterpret_cast<float*>(smem), acc, wm * WTM, wn * WTN, n0, etid, rowmap, ev);
    stamp();
  } else {
    const bool vec_ok = !p.store_nchw && (p.N & 3) == 0 && (p.ldc & 3) == 0 && (p.ldr & 3) == 0 &&
                        (p.ld_rowbias & 3) == 0 && (p.ld_aux & 3) == 0;
    if (vec_ok) {
      epilogue_block<BM, BN, NTE, TM, TN, EPI>(p, z, 0, reinterpret_cast<float*>(smem), acc, wm * WTM, wn * WTN, n0, etid, rowmap, ev);
      return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int rowv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) rowv[r] = rowmap(wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
      epilogue_band<TN>(p, z, 0, rowv, acc[i], n0 + wn * WTN, l31);
    }
  }
#endif
}

template <int BM, int BN, int WGM, int WGN, int NST, int MODE, int EPI, int NSTB = NST, int PS = 0>
hipError_t launch_cfg(const GemmParams& p, int zdim, hipStream_t stream) {
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  constexpr size_t ring = ((size_t)BM * NST + (size_t)BN * NSTB) * BK * 2;     // operand rings (A: NST slots, W: NSTB slots)
  constexpr size_t stage = (size_t)BM * (BN + 4) * 4 + (size_t)BM * 8 + (MODE == 3 ? (size_t)BM * 4 : 0);   // epilogue tile + (mean, rstd) row table (+ MODE 3 pixel table)
  const size_t base = ring > stage ? ring : stage;
  const size_t lds = base;
  if (p.ln_stats && p.ln_slots > 20) return hipErrorInvalidValue;     // LNS of the kernel
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  static bool attr_set = false;
  if (!attr_set) {
    const size_t cap = std::min<size_t>(160 * 1024, base + (size_t)5 * 64 * WGM * WGN * 8);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<BM, BN, WGM, WGN, NST, MODE, EPI, NSTB, PS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WGM, WGN, NST, MODE, EPI, NSTB, PS>), dim3(nbm * nbn, 1, MODE == 3 ? 4 * zdim : zdim),
                     dim3(64 * WGM * WGN * (1 + PS)), lds, stream, p);
  return hipGetLastError();
}


template <int BM, int BN, int WGM, int WGN, int NSTW, int EPI, int PS = 0>
hipError_t launch_halo(const GemmParams& pin, int zdim, hipStream_t stream) {
  constexpr int NT = 64 * WGM * WGN * (1 + PS), RPP = (PS ? 64 * WGM * WGN * PS : NT) / 8;
  GemmParams p = pin;
  if (!halo_patch(p.H, p.Wd, BM, &p.th, &p.tw)) return hipErrorInvalidValue;
  const int ppx = p.th * p.tw;
  if (ppx <= 0 || BM % ppx != 0 || p.H % p.th != 0 || p.Wd % p.tw != 0 || p.stride != 1 || p.ups != 0 || p.taps != 9)
    return hipErrorInvalidValue;
  const int PB = BM / ppx, HR = PB * (p.th + 2) * (p.tw + 2);
  const int APASS = (HR + RPP - 1) / RPP;
  if (APASS > 12) return hipErrorInvalidValue;
  if (p.Cin2 > 0 && (PS == 0 || 3 * BM > 2 * APASS * RPP || (p.Cin2 % BK) != 0 || !p.A2)) return hipErrorInvalidValue;   // folded-skip tail
  constexpr int WPASS = (BN + RPP - 1) / RPP;
  const size_t ring = ((size_t)2 * APASS * RPP + (size_t)NSTW * WPASS * RPP) * BK * 2;
  const size_t lds = ring + (size_t)std::max(BM, HR) * 4;          // + the prologue's halo-row table / the epilogue's row table
  p.halo_ring_bytes = (int)ring;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  static size_t attr = 0;
  if (lds > attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<BM, BN, WGM, WGN, NSTW, EPI, PS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr = lds;
  }
  const int npatch = p.M / ppx;
  const int nbm = (npatch + PB - 1) / PB, nbn = (p.N + BN - 1) / BN;
  if ((long)nbm * PB >= (1 << 22)) return hipErrorInvalidValue;      // FastDiv range of the patch decomposition
  hipLaunchKernelGGL((conv3x3_halo_kernel<BM, BN, WGM, WGN, NSTW, EPI, PS>), dim3(nbm * nbn, 1, zdim), dim3(NT), lds, stream, p);
  return hipGetLastError();
}

}  // namespace
