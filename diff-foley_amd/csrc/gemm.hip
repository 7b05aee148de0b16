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
#include "common.h"
#include "gemm.h"

namespace {

constexpr int BK = 64;

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

template <int N_>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

template <int BM, int BN, int WGM, int WGN, int NST, bool CONV>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)   // host pass only needs the launch stub (LDS-DMA builtins do not parse there)
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int AP = BM / 32, BP = BN / 32;
  static_assert(WGM * WGN == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* sA = reinterpret_cast<bf16_t*>(smem);
  bf16_t* sB = sA + NST * BM * BK;

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WGN, wn = wid % WGN;
  const int l31 = lane & 31, lh = lane >> 5;

  // ---- tile id with XCD-aware remap (block b runs on XCD b%8; give each XCD a contiguous range)
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int nblk = nbm * nbn;
  int lid;
  {
    const int bid = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (lid % nbm) * BM, n0 = (lid / nbm) * BN;

  int z = blockIdx.z;
  const int nk = p.K / BK;
  int kt0 = 0, kt1 = nk, batch = z;
  if (p.splitk > 1) {
    const int per = (nk + p.splitk - 1) / p.splitk;
    kt0 = z * per;
    kt1 = min(nk, kt0 + per);
    batch = 0;
  }
  const bf16_t* Ab = p.A + (long)batch * p.a_bs;
  const bf16_t* Wb = p.W + (long)batch * p.w_bs;

  // ---- per-thread staging coordinates: chunk c (8 bf16 = 16 B) of rows (tid>>3) + 32*i.
  // Operands are fetched with raw buffer loads: an out-of-range byte offset (OOB) makes the hardware return
  // zeros, so zero padding / ragged tiles need no branches and all loads of a K step issue back to back.
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, (int)p.w_bytes, 0x00020000);
  const int r0 = tid >> 3;
  const int c8 = ((tid & 7) ^ ((r0 >> 1) & 7)) * 8;   // source chunk that lands in LDS slot (tid&7) of row r0+32i
  unsigned a_off[AP];   // taps==1: byte offset of (row, chunk) ; taps==9: pixel index base n*H*W
  int a_iy[AP], a_ix[AP];
  const int UH = p.H << p.ups, UW = p.Wd << p.ups;
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int m = m0 + r0 + 32 * i;
    const bool mv = m < p.M;
    if (!CONV) {
      a_off[i] = mv ? (unsigned)(((long)m * p.lda + c8) * 2) : OOB;
      a_iy[i] = a_ix[i] = 0;
    } else {
      const int ohw = p.OH * p.OW;
      const int nb = m / ohw, rem = m - nb * ohw;
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      a_off[i] = (unsigned)(nb * p.H * p.Wd);
      a_iy[i] = mv ? oy * p.stride - 1 : -(1 << 20);
      a_ix[i] = ox * p.stride - 1;
    }
  }
  unsigned b_off[BP];
#pragma unroll
  for (int i = 0; i < BP; ++i) {
    const int n = n0 + r0 + 32 * i;
    b_off[i] = (n < p.N) ? (unsigned)(((long)n * p.K + c8) * 2) : OOB;
  }

  typedef __attribute__((address_space(3))) void* lds_ptr;
  constexpr int LPT = AP + BP;                  // DMA instructions per wave per K tile

  // Request tile T (relative to kt0) into ring slot ST: every wave writes 8 rows x 128 B (1 KiB, lane-linear)
  // per instruction.  Tiles past the end are requested with an out-of-bounds offset (zeros land in a dead slot),
  // so the DMA count per iteration is constant and the loop body has no branch.
  int d_tap = 0, d_cc = 0;       // conv: (tap, channel offset) of the NEXT tile to request, advanced incrementally
  if (CONV) {
    const int k0 = kt0 * BK;
    d_tap = k0 / p.Cin;
    d_cc = k0 - d_tap * p.Cin;
  }
#define DF_DMA(T, ST)                                                                             \
  {                                                                                             \
    bf16_t* a_ = sA + (ST) * BM * BK + wid * (8 * BK);                                          \
    bf16_t* b_ = sB + (ST) * BN * BK + wid * (8 * BK);                                          \
    const bool live = (T) < nt;                                                                 \
    const unsigned k0b = (unsigned)(kt0 + (T)) * (BK * 2);                                      \
    if (!CONV) {                                                                                \
      _Pragma("unroll") for (int i = 0; i < AP; ++i)                                            \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(a_ + i * 32 * BK), 16,          \
                                                 live ? a_off[i] + k0b : OOB, 0, 0, 0);         \
    } else {                                                                                    \
      const int ky = (d_tap * 11) >> 5, kx = d_tap - ky * 3;                                    \
      _Pragma("unroll") for (int i = 0; i < AP; ++i) {                                          \
        const int uy = a_iy[i] + ky, ux = a_ix[i] + kx;                                         \
        const bool v = live && ((unsigned)uy < (unsigned)UH) && ((unsigned)ux < (unsigned)UW);  \
        const int sy = uy >> p.ups, sx = ux >> p.ups;                                           \
        const unsigned off = ((a_off[i] + (unsigned)(sy * p.Wd + sx)) * (unsigned)p.lda + (unsigned)(d_cc + c8)) * 2u; \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(a_ + i * 32 * BK), 16, v ? off : OOB, 0, 0, 0); \
      }                                                                                         \
      d_cc += BK;                                                                               \
      if (d_cc == p.Cin) {                                                                      \
        d_cc = 0;                                                                               \
        ++d_tap;                                                                                \
      }                                                                                         \
    }                                                                                           \
    _Pragma("unroll") for (int i = 0; i < BP; ++i)                                              \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(b_ + i * 32 * BK), 16,            \
                                               live ? b_off[i] + k0b : OOB, 0, 0, 0);           \
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- prologue: fill NST-1 ring slots
  const int nt = kt1 - kt0;
#pragma unroll
  for (int t = 0; t < NST - 1; ++t) DF_DMA(t, t);

  // LDS fragment addresses (elements) of k-step 0; k-step s toggles the chunk index by 2*s
  int fa[TM], fb[TN], sa[TM], sb[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = wm * WTM + i * 32 + l31;
    fa[i] = row * BK;
    sa[i] = (row >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * WTN + j * 32 + l31;
    fb[j] = row * BK;
    sb[j] = (row >> 1) & 7;
  }
#define DF_FRAG(DSTA, DSTB, S)                                                                    \
  {                                                                                             \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                              \
      DSTA[i] = *reinterpret_cast<const bf16x8*>(a + fa[i] + (((2 * (S) + lh) ^ sa[i]) << 3));  \
    _Pragma("unroll") for (int j = 0; j < TN; ++j)                                              \
      DSTB[j] = *reinterpret_cast<const bf16x8*>(b + fb[j] + (((2 * (S) + lh) ^ sb[j]) << 3));  \
  }
#define DF_MMA(SRCA, SRCB)                                                                        \
  {                                                                                             \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                              \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                            \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(SRCA[i], SRCB[j], acc[i][j], 0, 0, 0); \
  }

  int st = 0, dst = NST - 1;
  for (int it = 0; it < nt; ++it) {
    // tile `it` has landed once at most (NST-2) younger tiles of this wave are still in flight
    wait_vmcnt<(NST - 2) * LPT>();
    __builtin_amdgcn_s_barrier();      // every wave's part of tile `it` visible; everyone is done with tile it-1
    const bf16_t* a = sA + st * BM * BK;
    const bf16_t* b = sB + st * BN * BK;
    bf16x8 a0[TM], b0[TN], a1[TM], b1[TN];
    DF_FRAG(a0, b0, 0);
    DF_DMA(it + NST - 1, dst);         // refill the slot tile it-1 used (overlaps the first MFMAs)
    DF_FRAG(a1, b1, 1);
    DF_MMA(a0, b0);
    DF_FRAG(a0, b0, 2);
    DF_MMA(a1, b1);
    DF_FRAG(a1, b1, 3);
    DF_MMA(a0, b0);
    DF_MMA(a1, b1);
    st = (st + 1 == NST) ? 0 : st + 1;
    dst = (dst + 1 == NST) ? 0 : dst + 1;
  }
  wait_vmcnt<0>();                     // dead-slot requests of the last iterations must land before LDS is released

  // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  // All loads of a 32x32 tile (bias, per-sample bias, residual) are issued unconditionally from clamped
  // addresses before any use, so they overlap instead of serialising behind per-element branches.
  if (p.splitk > 1) {
    float* part = p.partial + (long)z * p.M * p.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * WTN + j * 32 + l31;
        if (col >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row < p.M) part[(long)row * p.N + col] = acc[i][j][r];
        }
      }
    return;
  }
  const bool has_bias = p.bias != nullptr, has_rb = p.rowbias != nullptr, has_res = p.res != nullptr;
  const int nout = p.geglu ? (p.N >> 1) : p.N;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    int rowv[16], rowc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      rowv[r] = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      rowc[r] = min(rowv[r], p.M - 1);
    }
    constexpr int JS = 1;
#pragma unroll
    for (int j = 0; j < TN; j += JS) {
      if (p.geglu) {
        if constexpr (TN % 2 == 0) {
          if (j & 1) continue;
          const int xcol = n0 + wn * WTN + j * 32 + l31;
          const int xc = min(xcol, p.N - 33);
          const int ocol = (xcol >> 6) * 32 + (xcol & 63);
          const float bx = has_bias ? p.bias[xc] : 0.f, bg = has_bias ? p.bias[xc + 32] : 0.f;
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float xv = acc[i][j][r] * p.alpha + bx;
            const float gv = acc[i][j + 1][r] * p.alpha + bg;
            v[r] = xv * gelu_erf(gv);
          }
          if (has_res) {
            float rr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) rr[r] = p.res[(long)batch * p.res_bs + (long)rowc[r] * p.ldr + min(ocol, nout - 1)];
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
      const int col = n0 + wn * WTN + j * 32 + l31;
      const int cc = min(col, p.N - 1);
      const float b0 = has_bias ? p.bias[cc] : 0.f;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] * p.alpha + b0;
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
#endif
}

// Sums the split-K partial slabs and applies the same epilogue.  One thread per output element.
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

template <int BM, int BN, int WGM, int WGN, int NST, bool CONV>
hipError_t launch_cfg(const GemmParams& p, int zdim, hipStream_t stream) {
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const size_t lds = (size_t)(BM + BN) * BK * 2 * NST;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<BM, BN, WGM, WGN, NST, CONV>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WGM, WGN, NST, CONV>), dim3(nbm * nbn, 1, zdim), dim3(256), lds, stream, p);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm(const GemmParams& p, int tile_cfg, int batch, hipStream_t stream) {
  const int zdim = (p.splitk > 1) ? p.splitk : (batch > 0 ? batch : 1);
  hipError_t e;
  const bool conv = p.taps == 9;
#define DF_CASE(T, BM, BN, WGM, WGN, NST)                                              \
  case T:                                                                             \
    e = conv ? launch_cfg<BM, BN, WGM, WGN, NST, true>(p, zdim, stream)               \
             : launch_cfg<BM, BN, WGM, WGN, NST, false>(p, zdim, stream);             \
    break;
  switch (tile_cfg) {
    DF_CASE(TILE_128x128, 128, 128, 2, 2, 3)
    DF_CASE(TILE_128x64, 128, 64, 2, 2, 4)
    DF_CASE(TILE_64x128, 64, 128, 2, 2, 4)
    DF_CASE(TILE_64x64, 64, 64, 2, 2, 4)
    DF_CASE(TILE_32x128, 32, 128, 1, 4, 4)
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess) return e;
  if (p.splitk > 1) {
    const int nout = p.geglu ? (p.N >> 1) : p.N;
    const long total = (long)p.M * nout;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p);
    e = hipGetLastError();
  }
  return e;
}
