// MFMA bf16 implicit-GEMM for gfx950: conv3x3 / conv1x1 / Linear / batched matmul with fused epilogues.
//
// Block = 256 threads = 4 wavefronts (64 lanes) in a WGM x WGN grid; each wavefront owns a
// (BM/WGM) x (BN/WGN) output tile made of 32x32 MFMA tiles (v_mfma_f32_32x32x16_bf16).
// K is walked in steps of 64.  Operand tiles are staged global -> VGPR -> LDS with the next
// tile's global loads issued before the current tile's MFMAs (T14 split), LDS double buffered,
// one barrier per K step.  LDS rows are 128 B (64 bf16); the 16-B chunk index is XOR-swizzled
// with (row>>1)&7 so the ds_read_b128 fragment reads of a 16-lane group hit 16 distinct slots.
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

__device__ __forceinline__ void epi_out(const GemmParams& p, int z, int row, int col, float v) {
  if (p.res) v += p.res[(long)z * p.res_bs + (long)row * p.ldr + col];
  long idx;
  if (p.store_nchw) {
    const int b = row / p.hw_out, px = row - b * p.hw_out;
    const int nout = p.geglu ? (p.N >> 1) : p.N;
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

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmParams p) {
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int AP = BM / 32, BP = BN / 32;
  static_assert(WGM * WGN == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* sA = reinterpret_cast<bf16_t*>(smem);
  bf16_t* sB = sA + 2 * BM * BK;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
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

  // ---- per-thread staging coordinates: chunk c (8 bf16 = 16 B) of rows (tid>>3) + 32*i
  const int c8 = (tid & 7) * 8;
  const int r0 = tid >> 3;
  long a_off[AP];   // taps==1: element offset of the row; taps==9: pixel base (n*H*W)
  int a_iy[AP], a_ix[AP];
  const int UH = p.H << p.ups, UW = p.Wd << p.ups;
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int m = m0 + r0 + 32 * i;
    const bool mv = m < p.M;
    if (p.taps == 1) {
      a_off[i] = mv ? (long)m * p.lda : -1;
      a_iy[i] = a_ix[i] = 0;
    } else {
      const int ohw = p.OH * p.OW;
      const int nb = m / ohw, rem = m - nb * ohw;
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      a_off[i] = (long)nb * p.H * p.Wd;
      a_iy[i] = mv ? oy * p.stride - 1 : -(1 << 20);
      a_ix[i] = ox * p.stride - 1;
    }
  }
  long b_off[BP];
#pragma unroll
  for (int i = 0; i < BP; ++i) {
    const int n = n0 + r0 + 32 * i;
    b_off[i] = (n < p.N) ? (long)n * p.K : -1;
  }

  uint4 ra[AP], rb[BP];
  const uint4 zero4 = make_uint4(0, 0, 0, 0);

#define DF_GLOAD(KT)                                                                              \
  {                                                                                             \
    const int k0 = (KT) * BK;                                                                   \
    if (p.taps == 1) {                                                                          \
      _Pragma("unroll") for (int i = 0; i < AP; ++i) {                                          \
        ra[i] = zero4;                                                                          \
        if (a_off[i] >= 0) ra[i] = *reinterpret_cast<const uint4*>(Ab + a_off[i] + k0 + c8);    \
      }                                                                                         \
    } else {                                                                                    \
      const int tap = k0 / p.Cin, cc = k0 - tap * p.Cin;                                        \
      const int ky = tap / 3, kx = tap - ky * 3;                                                \
      _Pragma("unroll") for (int i = 0; i < AP; ++i) {                                          \
        const int uy = a_iy[i] + ky, ux = a_ix[i] + kx;                                         \
        const bool v = ((unsigned)uy < (unsigned)UH) && ((unsigned)ux < (unsigned)UW);          \
        const int sy = uy >> p.ups, sx = ux >> p.ups;                                           \
        const long off = (a_off[i] + (long)sy * p.Wd + sx) * p.lda + cc + c8;                   \
        ra[i] = zero4;                                                                          \
        if (v) ra[i] = *reinterpret_cast<const uint4*>(Ab + off);                               \
      }                                                                                         \
    }                                                                                           \
    _Pragma("unroll") for (int i = 0; i < BP; ++i) {                                            \
      rb[i] = zero4;                                                                            \
      if (b_off[i] >= 0) rb[i] = *reinterpret_cast<const uint4*>(Wb + b_off[i] + k0 + c8);      \
    }                                                                                           \
  }
#define DF_SSTORE(BUF)                                                                            \
  {                                                                                             \
    bf16_t* a_ = sA + (BUF) * BM * BK;                                                          \
    bf16_t* b_ = sB + (BUF) * BN * BK;                                                          \
    const int c_ = tid & 7;                                                                     \
    _Pragma("unroll") for (int i = 0; i < AP; ++i) {                                            \
      const int r = r0 + 32 * i;                                                                \
      *reinterpret_cast<uint4*>(a_ + r * BK + ((c_ ^ ((r >> 1) & 7)) << 3)) = ra[i];            \
    }                                                                                           \
    _Pragma("unroll") for (int i = 0; i < BP; ++i) {                                            \
      const int r = r0 + 32 * i;                                                                \
      *reinterpret_cast<uint4*>(b_ + r * BK + ((c_ ^ ((r >> 1) & 7)) << 3)) = rb[i];            \
    }                                                                                           \
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (kt0 < kt1) {
    DF_GLOAD(kt0);
    DF_SSTORE(0);
  }
  __syncthreads();
  int buf = 0;
  for (int kt = kt0; kt < kt1; ++kt) {
    const bool more = (kt + 1 < kt1);
    if (more) DF_GLOAD(kt + 1);
    const bf16_t* a = sA + buf * BM * BK;
    const bf16_t* b = sB + buf * BN * BK;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8 af[TM], bfr[TN];
      const int ch = 2 * s + lh;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * WTM + i * 32 + l31;
        af[i] = *reinterpret_cast<const bf16x8*>(a + row * BK + ((ch ^ ((row >> 1) & 7)) << 3));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * WTN + j * 32 + l31;
        bfr[j] = *reinterpret_cast<const bf16x8*>(b + row * BK + ((ch ^ ((row >> 1) & 7)) << 3));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (more) DF_SSTORE(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
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
  if (p.geglu) {
    if constexpr (TN % 2 == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; j += 2) {
          const int xcol = n0 + wn * WTN + j * 32 + l31;
          if (xcol >= p.N) continue;
          const int ocol = (xcol >> 6) * 32 + (xcol & 63);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row >= p.M) continue;
            const float xv = epi_bias(p, row, xcol, acc[i][j][r]);
            const float gv = epi_bias(p, row, xcol + 32, acc[i][j + 1][r]);
            epi_out(p, batch, row, ocol, xv * gelu_erf(gv));
          }
        }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * WTN + j * 32 + l31;
      if (col >= p.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < p.M) epi_out(p, batch, row, col, epi_bias(p, row, col, acc[i][j][r]));
      }
    }
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

template <int BM, int BN, int WGM, int WGN>
hipError_t launch_cfg(const GemmParams& p, int zdim, hipStream_t stream) {
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const size_t lds = (size_t)(BM + BN) * BK * 2 * 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<BM, BN, WGM, WGN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WGM, WGN>), dim3(nbm * nbn, 1, zdim), dim3(256), lds, stream, p);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm(const GemmParams& p, int tile_cfg, int batch, hipStream_t stream) {
  const int zdim = (p.splitk > 1) ? p.splitk : (batch > 0 ? batch : 1);
  hipError_t e;
  switch (tile_cfg) {
    case TILE_128x128: e = launch_cfg<128, 128, 2, 2>(p, zdim, stream); break;
    case TILE_128x64:  e = launch_cfg<128, 64, 2, 2>(p, zdim, stream); break;
    case TILE_64x128:  e = launch_cfg<64, 128, 2, 2>(p, zdim, stream); break;
    case TILE_64x64:   e = launch_cfg<64, 64, 2, 2>(p, zdim, stream); break;
    case TILE_32x128:  e = launch_cfg<32, 128, 1, 4>(p, zdim, stream); break;
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
