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
#include <stdlib.h>

#include "gemm_impl.h"


hipError_t launch_gemm_m0a(int tile_cfg, int epi, const GemmParams& p, int zdim, hipStream_t stream);
hipError_t launch_gemm_m0b(int tile_cfg, int epi, const GemmParams& p, int zdim, hipStream_t stream);
hipError_t launch_gemm_m1(int tile_cfg, int epi, const GemmParams& p, int zdim, hipStream_t stream);
hipError_t launch_gemm_m2(int tile_cfg, int epi, const GemmParams& p, int zdim, hipStream_t stream);
hipError_t launch_gemm_halo(int tile_cfg, int epi, const GemmParams& p, int zdim, hipStream_t stream);
hipError_t launch_gemm_m3(int tile_cfg, int epi, const GemmParams& p, int zdim, hipStream_t stream);
hipError_t launch_gemm_ps(int mode, int tile_cfg, int epi, const GemmParams& p, int zdim, hipStream_t stream);
hipError_t launch_gemm_ps_small(int mode, int tile_cfg, int epi, const GemmParams& p, int zdim, hipStream_t stream);
hipError_t launch_gemm_pgeglu(int tile_cfg, const GemmParams& p, hipStream_t stream);
bool pgeglu_valid(const GemmParams& p, int tile, int batch, int splitk);
hipError_t launch_gemm_wgeglu(int tile_cfg, const GemmParams& p, hipStream_t stream);
bool wgeglu_valid(const GemmParams& p, int tile, int batch, int splitk);

namespace {

// Sums the split-K partial slabs and applies the same epilogue.  One thread per output element (scalar fallback:
// GEGLU, NCHW stores, unaligned leading dimensions).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmParams p) {
  const int nout = p.geglu ? (p.N >> 1) : p.N;
  const long total = (long)p.M * nout;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int row = (int)(e / nout), oc = (int)(e - (long)row * nout);
    if (p.geglu) {
      const int xcol = (oc >> 5) * 64 + (oc & 31);
      float xs = 0.f, gs = 0.f;
      for (int s = 0; s < p.splitk; ++s) {
        const float* part = p.partial + ((long)s * p.M + row) * p.N;
        xs += part[xcol];
        gs += part[xcol + 32];
      }
      epi_out(p, 0, row, oc, epi_bias(p, row, xcol, xs) * gelu_erf(epi_bias(p, row, xcol + 32, gs)));
    } else {
      float v = 0.f;
      for (int s = 0; s < p.splitk; ++s) v += p.partial[((long)s * p.M + row) * p.N + oc];
      epi_out(p, 0, row, oc, epi_bias(p, row, oc, v));
    }
  }
}

// Split-K reduce of UNetModel.out on a classifier-free-guidance batch (round 5): rows [0, M/2) are the unconditional half, rows
// [M/2, M) the conditional one; both are reduced in slab order, get alpha / bias like the scalar reduce above, and the guided
// eps  e_u + scale (e_c - e_u)  leaves as NCHW -- the arithmetic of splitk_reduce_kernel followed by cfg_combine_kernel, bit for bit,
// in one launch instead of two.
__global__ __launch_bounds__(256) void splitk_reduce_cfg_kernel(GemmParams p) {
  const int half = p.M >> 1;
  const long total = (long)half * p.N;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int row = (int)(e / p.N), oc = (int)(e - (long)row * p.N);
    float vu = 0.f, vc = 0.f;
    for (int s = 0; s < p.splitk; ++s) {
      vu += p.partial[((long)s * p.M + row) * p.N + oc];
      vc += p.partial[((long)s * p.M + row + half) * p.N + oc];
    }
    const float u = epi_bias(p, row, oc, vu), c = epi_bias(p, row + half, oc, vc);
    const int b = row / p.hw_out, px = row - b * p.hw_out;
    p.cfg_out[((long)b * p.N + oc) * p.hw_out + px] = u + p.cfg_scale * (c - u);
  }
}

// Vectorised reduce: one thread per 4 consecutive output columns; the SK slab loads of a thread are independent and
// issued together (compile-time split count), then alpha / bias / FiLM bias / residual / ReLU and one 16-B (fp32) or
// 8-B (operand type) store.  Summation order s = 0..SK-1 is fixed, so results are deterministic.
template <int SK>
__global__ __launch_bounds__(256) void splitk_reduce_vec_kernel(GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  const DfTouch ka = gemm_kernarg_touch();      // kernel-argument lines (and the code behind the pc) into L2 beside the first scalar loads
#endif
  const int n4 = p.N >> 2;
  const long total = (long)p.M * n4;
  const long slab = (long)p.M * p.N;
  const bool has_bias = p.bias != nullptr, has_rb = p.rowbias != nullptr, has_res = p.res != nullptr;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int row = (int)(e / n4), col = (int)(e - (long)row * n4) * 4;
    const float* src = p.partial + (long)row * p.N + col;
    float4 t[SK];
#pragma unroll
    for (int s = 0; s < SK; ++s) t[s] = *reinterpret_cast<const float4*>(src + s * slab);
    float4 v = t[0];
#pragma unroll
    for (int s = 1; s < SK; ++s) {
      v.x += t[s].x; v.y += t[s].y; v.z += t[s].z; v.w += t[s].w;
    }
    v.x *= p.alpha; v.y *= p.alpha; v.z *= p.alpha; v.w *= p.alpha;
    if (p.ln_stats) {        // LayerNorm folded into the GEMM: row statistics from the producer's per-slot partials
      const float2* sp = p.ln_stats + (long)row * p.ln_slots;
      float s1 = 0.f, s2 = 0.f;
      for (int i = 0; i < p.ln_slots; ++i) {
        const float2 t2 = sp[i];
        s1 += t2.x;
        s2 += t2.y;
      }
      const float inv = 1.0f / (float)p.ln_C;
      const float mean = s1 * inv, rstd = rsqrtf(fmaxf(s2 * inv - mean * mean, 0.f) + p.ln_eps);
      const float4 cs = *reinterpret_cast<const float4*>(&p.ln_cs[col]);
      v.x = rstd * (v.x - mean * cs.x); v.y = rstd * (v.y - mean * cs.y);
      v.z = rstd * (v.z - mean * cs.z); v.w = rstd * (v.w - mean * cs.w);
    }
    if (has_bias) {
      const float4 b = *reinterpret_cast<const float4*>(&p.bias[col]);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (has_rb) {
      const int ri = (p.rowbias_mode == 1) ? (row / p.rows_per_sample) : (row % p.rows_per_sample);
      const float4 b = *reinterpret_cast<const float4*>(&p.rowbias[(long)ri * p.ld_rowbias + col]);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (has_res) {
      const float4 b = *reinterpret_cast<const float4*>(&p.res[(long)row * p.ldr + col]);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (p.silu && !p.ln_stats && !p.stats) {      // same rule as the in-kernel epilogue (gemm_impl.h DF_EPI_LOOP)
      v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w);
    }
    if (p.relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    if (p.stats) {           // N % 64 == 0: 16 consecutive threads hold one 64-column slot of one row
      const float s1 = row16_sum((v.x + v.y) + (v.z + v.w));
      const float s2 = row16_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
      if ((threadIdx.x & 15) == 0) {
        p.stats[(long)row * p.stats_slots + (col >> 6)] = make_float2(s1, s2);
        if (p.dup_rows) p.stats[(long)(row + p.dup_rows) * p.stats_slots + (col >> 6)] = make_float2(s1, s2);
      }
    }
    for (int rep = 0; rep < (p.dup_rows ? 2 : 1); ++rep) {      // CFG prefix: every row is stored for both halves of the batch
      const long orow = row + (rep ? p.dup_rows : 0);
      const long idx = orow * p.ldc + col;
      if (p.no_c_store) {       // producer whose fp32 value nobody reads (operand copy + statistics only)
      } else if (p.out_bf16)
        st_wt(reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + idx), make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)));
      else
        st_wt(reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + idx), v);
      if (p.aux) st_wt(reinterpret_cast<uint2*>(p.aux + orow * p.ld_aux + col), make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)));
    }
  }
#if defined(__HIP_DEVICE_COMPILE__)
  gemm_kernarg_touch_end(ka);
#endif
}

}  // namespace

bool gemm_tile_valid(const GemmParams& p, int tile, int batch, int splitk) {
  const int nk = p.K / 64;
  if (p.taps == 4) {     // phase-decomposed upsample conv (MODE 3): a subset of the generic tiles, plain epilogues only
    static const bool ok3[TILE_ALL] = {false, false, false, true, false, false, false, false, true, true, true, true, true, true, false, false, false, false,
                                       false, false, false, false, false, false, false, false, false, false, false, false, false, false, false, false, false};     // (no producer-specialised MODE 3 instantiation)
    if (tile < 0 || tile >= TILE_ALL || !ok3[tile] || batch > 1) return false;
    if (p.geglu || p.vt || p.ln_stats || p.stats || p.w_rows > 0 || p.sm_w > 0 || p.Cin2 > 0 || p.res || p.store_nchw) return false;
    if (splitk > 1 && (p.N & 3) != 0) return false;
    return splitk == 1 || nk / splitk >= 2;
  }
  if (gemm_tile_is_pgeglu(tile)) return pgeglu_valid(p, tile, batch, splitk);
  if (gemm_tile_is_wgeglu(tile)) return wgeglu_valid(p, tile, batch, splitk);
  if (gemm_tile_is_ps(tile) && p.taps == 9 && (p.stride != 1 || p.ups)) return false;     // instantiated for MODE 0 and 1
  if (!gemm_tile_is_halo(tile)) {
    if (tile < 0 || tile >= TILE_ALL) return false;
    if (p.geglu && ((p.N & 63) != 0 || (p.ldc & 3) != 0)) return false;   // GEGLU needs the vectorised block epilogue
    int bm_, bn_;
    gemm_tile_dims(tile, &bm_, &bn_);
    if (p.vt && (splitk > 1 || batch > 1 || p.vt_col0 % bn_ != 0)) return false;   // transposed-V tiles are whole tiles
    if (p.ln_stats && (batch > 1 || (p.geglu && splitk > 1))) return false;
    if (p.ln_stats && p.ln_slots > gemm_ln_max_slots()) return false;    // thread r folds the partials of tile row r (<= 20 slots)
    if (p.stats && batch > 1) return false;
    if (p.w_rows > 0 && (batch > 1 || p.taps != 1 || p.w_rows % bm_ != 0 || p.M % p.w_rows != 0)) return false;
    if (p.sm_w > 0 && (p.sm_w != 32 || splitk > 1 || !p.ln_stats || !p.out_bf16 || p.w_rows <= 0 || (p.N & 31) != 0 || p.geglu ||
                       p.vt || p.res || p.rowbias || p.aux || p.alpha != 1.f || !p.bias))
      return false;
    if (p.dup_rows > 0 && (batch > 1 || p.geglu || p.vt || p.sm_w > 0 || p.store_nchw || (p.N & 3) != 0 || (p.ldc & 3) != 0 ||
                           (p.ldr & 3) != 0 || (p.ld_rowbias & 3) != 0 || (p.ld_aux & 3) != 0))
      return false;
    if (p.Cin2 > 0 && (batch > 1 || (p.Cin2 & 63) != 0 || !p.A2)) return false;
    if (p.Cin2 > 0 && p.taps == 9 && (p.stride != 1 || p.ups)) return false;    // conv: the folded 1x1 skip connection
    if (p.Cin2 > 0 && p.taps != 9 && (p.taps != 1 || p.Cin2 >= p.K)) return false;   // linear: K columns [K-Cin2, K) from A2
    if (splitk > 1 && (p.N & 3) != 0) return false;          // partial slabs are written and reduced as float4
    return splitk == 1 || (batch <= 1 && nk / splitk >= 2);
  }
  if (splitk > 1 && (p.N & 3) != 0) return false;
  if (p.dup_rows > 0 && (p.store_nchw || (p.N & 3) != 0 || (p.ldc & 3) != 0 || (p.ldr & 3) != 0 || (p.ld_rowbias & 3) != 0 || (p.ld_aux & 3) != 0))
    return false;
  // the folded skip connection: generic stride-1 kernel, and (round 5) a one-tap K tail of the producer-specialised halo tiles
  const bool halo_ps = tile >= TILE_HALO_PS_192x64 && tile <= TILE_HALO_PS_128x128;
  constexpr bool no_halo_skip = false;
  if (p.Cin2 > 0 && (!halo_ps || no_halo_skip || (p.Cin2 & 63) != 0 || !p.A2 || (p.lda2 & 7) != 0)) return false;
  if (p.taps != 9 || p.stride != 1 || p.ups != 0 || p.geglu || batch > 1) return false;
  int bm, bn, th, tw;
  gemm_tile_dims(tile, &bm, &bn);
  if (!halo_patch(p.H, p.Wd, bm, &th, &tw)) return false;
  const int threads = gemm_halo_dma_threads(tile), rpp = threads / 8;
  const int hr = (bm / (th * tw)) * (th + 2) * (tw + 2);
  const int apass = (hr + rpp - 1) / rpp, wpass = (bn + rpp - 1) / rpp;
  if (apass > 12) return false;
  if (p.Cin2 > 0 && 3 * bm > 2 * apass * rpp) return false;      // the tail's three activation slots live in the two halo buffers
  const int nstw = gemm_halo_ring(tile);   // weight ring depth (4; 8 for the weight-streaming variants)
  if (((size_t)2 * apass * rpp + (size_t)nstw * wpass * rpp) * 128 + (size_t)std::max(bm, hr) * 4 > 160 * 1024) return false;
  if ((p.lda & 7) != 0) return false;           // the halo-row table keeps a 3-bit key in the low bits of a pixel's byte offset
  const int nchunk = p.Cin / 64;
  return splitk == 1 || nchunk / splitk >= 1;
}

hipError_t launch_gemm(const GemmParams& p, int tile_cfg, int batch, hipStream_t stream) {
  if (gemm_tile_is_pgeglu(tile_cfg)) {
    if (!pgeglu_valid(p, tile_cfg, batch, p.splitk)) return hipErrorInvalidValue;
    return launch_gemm_pgeglu(tile_cfg, p, stream);
  }
  if (gemm_tile_is_wgeglu(tile_cfg)) {
    if (!wgeglu_valid(p, tile_cfg, batch, p.splitk)) return hipErrorInvalidValue;
    return launch_gemm_wgeglu(tile_cfg, p, stream);
  }
  const int zdim = (p.splitk > 1) ? p.splitk : (batch > 0 ? batch : 1);
  if (p.ln_stats || p.stats || p.vt) {     // these epilogues exist in the vectorised paths only
    const bool vec = !p.store_nchw && (p.N & 63) == 0 && (p.ldc & 3) == 0 && (p.ldr & 3) == 0 && (p.ld_rowbias & 3) == 0 &&
                     (p.ld_aux & 3) == 0;
    if (!vec || (p.vt && ((p.M & 3) != 0 || (p.vt_T & 3) != 0 || (p.ldvt & 3) != 0 || p.taps != 1))) return hipErrorInvalidValue;
    if (p.ln_stats && p.taps != 1) return hipErrorInvalidValue;
  }
  // epilogue specialisation (gemm_impl.h): the kernel carries only the code it runs
  const bool vec = !p.store_nchw && (p.N & 3) == 0 && (p.ldc & 3) == 0 && (p.ldr & 3) == 0 && (p.ld_rowbias & 3) == 0 &&
                   (p.res_bs & 3) == 0 && (p.c_bs & 3) == 0 && (p.ld_aux & 3) == 0;
  int epi;
  if (p.splitk > 1) epi = EPI_SPLITK;
  else if (p.sm_w > 0) epi = EPI_XS;
  else if (p.geglu) epi = EPI_GEGLU;
  else if (p.ln_stats || p.vt) epi = EPI_LNC;
  else if (p.stats) epi = EPI_PROD;
  else if (!vec || p.relu || p.aux || p.alpha != 1.f) epi = EPI_ANY;
  else epi = EPI_LEAN;
  if ((epi == EPI_GEGLU || epi == EPI_LNC || epi == EPI_PROD) && (p.taps != 1 || p.alpha != 1.f || p.relu || p.silu || !vec))
    return hipErrorInvalidValue;
  if (epi == EPI_LNC && (!p.ln_stats || p.aux)) return hipErrorInvalidValue;
  if (epi == EPI_XS && (!p.ln_stats || p.taps != 1 || !vec || (p.N & 63) != 0)) return hipErrorInvalidValue;
  hipError_t e;
  // MODE 0: linear / 1x1;  1: 3x3 stride 1 (tap offsets are linear, 2 VALU per request);  2: 3x3 stride 2 / upsampled
  const int mode = (p.taps == 4) ? 3 : (p.taps != 9) ? 0 : ((p.stride == 1 && !p.ups) ? 1 : 2);
  if (mode == 3) {
    if (batch > 1 || gemm_tile_is_halo(tile_cfg) || p.OH != p.H || p.OW != p.Wd || p.K != 4 * p.Cin || p.w_bs != (long)p.N * p.K)
      return hipErrorInvalidValue;
    e = launch_gemm_m3(tile_cfg, epi, p, zdim, stream);
  } else if (gemm_tile_is_halo(tile_cfg)) e = launch_gemm_halo(tile_cfg, epi, p, zdim, stream);
  else if (gemm_tile_is_ps(tile_cfg)) e = (tile_cfg >= TILE_PS_64x64) ? launch_gemm_ps_small(mode, tile_cfg, epi, p, zdim, stream) : launch_gemm_ps(mode, tile_cfg, epi, p, zdim, stream);
  else if (mode == 0) e = (tile_cfg <= TILE_256x128) ? launch_gemm_m0a(tile_cfg, epi, p, zdim, stream) : launch_gemm_m0b(tile_cfg, epi, p, zdim, stream);
  else if (mode == 1) e = launch_gemm_m1(tile_cfg, epi, p, zdim, stream);
  else e = launch_gemm_m2(tile_cfg, epi, p, zdim, stream);
  if (e != hipSuccess) return e;
  if (p.splitk > 1 && !p.defer_reduce && mode == 3) {
    GemmParams q = p;
    q.M = 4 * p.M;           // the slabs hold the x2 output map
    q.taps = 1;
    const long total = (long)q.M * (q.N >> 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if ((q.N & 3) != 0 || (q.ldc & 3) != 0 || q.store_nchw) return hipErrorInvalidValue;
#define DF_RED3(SK) case SK: hipLaunchKernelGGL(splitk_reduce_vec_kernel<SK>, dim3(blocks), dim3(256), 0, stream, q); break;
    switch (p.splitk) {
      DF_RED3(2) DF_RED3(3) DF_RED3(4) DF_RED3(6) DF_RED3(8) DF_RED3(12) DF_RED3(16) DF_RED3(24) DF_RED3(32)
      default: hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, q); break;
    }
#undef DF_RED3
    return hipGetLastError();
  }
  if (p.splitk > 1 && p.dup_rows > 0 && p.defer_reduce) return hipErrorInvalidValue;
  if (p.splitk > 1 && !p.defer_reduce) {
    const bool vec = !p.geglu && !p.store_nchw && (p.N & 3) == 0 && (p.ldc & 3) == 0 && (p.ldr & 3) == 0 &&
                     (p.ld_rowbias & 3) == 0 && (p.ld_aux & 3) == 0;
    if (vec) {
      const long total = (long)p.M * (p.N >> 2);
      int blocks = (int)((total + 255) / 256);
      if (blocks > 4096) blocks = 4096;
#define DF_RED(SK) case SK: hipLaunchKernelGGL(splitk_reduce_vec_kernel<SK>, dim3(blocks), dim3(256), 0, stream, p); break;
      switch (p.splitk) {
        DF_RED(2) DF_RED(3) DF_RED(4) DF_RED(6) DF_RED(8) DF_RED(12) DF_RED(16) DF_RED(24) DF_RED(32)
        default: hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p); break;
      }
#undef DF_RED
    } else if (p.cfg_out) {
      if (!p.store_nchw || p.geglu || p.res || p.aux || p.silu || p.relu || (p.M & 1) || p.hw_out <= 0 || (p.M >> 1) % p.hw_out != 0)
        return hipErrorInvalidValue;
      const long total = (long)(p.M >> 1) * p.N;
      int blocks = (int)((total + 255) / 256);
      if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(splitk_reduce_cfg_kernel, dim3(blocks), dim3(256), 0, stream, p);
    } else {
      const int nout = p.geglu ? (p.N >> 1) : p.N;
      const long total = (long)p.M * nout;
      int blocks = (int)((total + 255) / 256);
      if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p);
    }
    e = hipGetLastError();
  }
  return e;
}
