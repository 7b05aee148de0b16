// Fused softmax(Q K^T * scale) V on MFMA for gfx950 (UNet self-attention, tokens 16..1024, and
// cross-attention over 32/33 CAVP context tokens).  Scores never touch HBM.
//
// Work split: block = NW wavefronts, each wavefront owns 32 query rows of one (sample, head);
// keys are walked in tiles of 32 staged in LDS (K tile row-major, V tile already transposed by
// the producer GEMM, so both stage with plain 16-B copies).
//
// MFMA formulation (v_mfma_f32_32x32x16_bf16, "swapped" so softmax stays lane-local):
//   S^T[key][q]  = sum_d K[key][d] Q[q][d]          A = K tile (LDS), B = Q (registers, loaded once)
//   O^T[d][q]   += sum_key V^T[d][key] P^T[key][q]  A = V^T tile (LDS), B = P (registers, straight from S^T)
// The C layout of S^T gives lane (q = lane&31, h = lane>>5) the 16 keys {(r&3) + 8(r>>2) + 4h}.  The PV
// contraction index is relabelled so that operand slot p of half h in 16-key chunk c means key
// (p&3) + 4h + 8(p>>2) + 16c: then P's registers 8c..8c+7 ARE the B fragment (no cross-lane traffic)
// and the A fragment is two 8-byte LDS reads of V^T (keys 16c+4h..+3 and 16c+8+4h..+3).
// Softmax statistics (running max m, running sum l) are fp32; one __shfl_xor(32) joins the two
// half-waves that share a query.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

namespace {

// KS = 32-key sub-tiles per loop iteration (round 5).  With KS = 1 a wavefront's iteration is ONE dependent chain -- K fragment read,
// 3-10 MFMAs into one accumulator, a 16-deep max chain, exponentials, a sum chain, the P V MFMAs, one block barrier -- ~2.2 k cycles
// for 7 MFMAs at D = 40 with two wavefronts per SIMD to hide it (34 us for the 1024-token self-attention, MFMA pipe busy 21 %).
// KS = 2 walks 64 keys per iteration as two INDEPENDENT sub-tiles (two score accumulators, two max / sum chains that meet once),
// one barrier and one staging round per 64 keys.  Used when Tk is a multiple of 64 and the two 64-key buffers fit (D <= 80).
template <int KS> struct AttnGeo {
  static constexpr int KT = 32 * KS;            // keys per iteration
  static constexpr int VROW = KT + 4;           // V^T LDS row stride in bf16 (72 B / 136 B: conflict-free ds_read_b64, 8-B aligned)
};

template <int D, int NW, int KS = 1>
__global__ __launch_bounds__(NW * 64) void attention_kernel(const bf16_t* __restrict__ Q, int ldq,
                                                            const bf16_t* __restrict__ K, int ldk,
                                                            const bf16_t* __restrict__ Vt, int ldvt, int heads, int Tq,
                                                            int Tk, float scale_log2e, bf16_t* __restrict__ O, int ldo) {
  // (argument order: everything the Q / K / V fetches need sits inside the 16 dwords preloaded into SGPRs at wavefront launch,
  // build.sh; the output pointer, needed last, arrives by s_load)
  constexpr int DKC = (D + 15) / 16;      // 16-wide contraction chunks for Q K^T
  constexpr int DKP = DKC * 16;           // padded head dim
  constexpr int DT = (D + 31) / 32;       // 32-row tiles of O^T
  constexpr int KROW = DKP + 8;           // K LDS row stride (bf16): +16 B pad de-conflicts ds_read_b128
  constexpr int KT = AttnGeo<KS>::KT, VROW = AttnGeo<KS>::VROW;
  // Row sum by the matrix pipe (round 5): when D is not a multiple of 32 the last 32-row tile of O^T has padding rows; row D of the
  // staged V^T tile is kept at 1.0, so O^T[D][q] accumulates sum_key P[key][q] beside the outputs -- rescaled with them, in fp32, over
  // the operand-rounded probabilities the numerator sums -- and the 16 adds per 32 keys of the lane-local sum chain disappear
  // (D = 40: 34 of 136 VALU instructions per 64 keys; the kernel is bound by its VALU issue, section 3).
  constexpr bool ONES = (D % 32) != 0;
  constexpr int ONE_T = D / 32, ONE_R = ((D % 32) / 8) * 4 + (D % 4);      // tile / accumulator register of row D (held by half-wave (D % 8) / 4)
  static_assert(!ONES || (D % 8) == 0, "row D sits in half-wave 0");
  __shared__ __attribute__((aligned(16))) bf16_t sK[2 * KT * KROW];
  __shared__ __attribute__((aligned(16))) bf16_t sV[2 * DT * 32 * VROW];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  // grid = (q-tiles, heads, samples).  (A head-major grid that pins each head's K/V to one XCD measured 25 % slower.)
  const int n = blockIdx.z, h = blockIdx.y;
  const int q0 = (blockIdx.x * NW + wid) * 32;
  const int q = q0 + l31;
  const bool qv = q < Tq;

  // ---- Q fragment (B operand of S^T): lane (q, lh) holds Q[q][16c + 8*lh .. +8], zero beyond D / Tq
  bf16x8 qf[DKC];
  {
    const bf16_t* qr = Q + ((long)n * Tq + (qv ? q : 0)) * ldq + h * D;
#pragma unroll
    for (int c = 0; c < DKC; ++c) {
      const int d0 = 16 * c + 8 * lh;
      const bool ok = qv && d0 < D;                                    // D % 8 == 0
      uint4 v = *reinterpret_cast<const uint4*>(qr + (ok ? d0 : 0));   // unconditional load, then select
      if (!ok) v = make_uint4(0, 0, 0, 0);
      qf[c] = *reinterpret_cast<bf16x8*>(&v);
    }
  }

  f32x16 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const bf16_t* Kb = K + (long)n * Tk * ldk + h * D;
  const bf16_t* Vb = Vt + ((long)n * heads + h) * D * ldvt;
  const int ntiles = (Tk + KT - 1) / KT;

  // ---- K / V^T tile staging, software pipelined (T14): the global loads of tile kt+1 are issued before the MFMAs
  // of tile kt and written to the other LDS buffer afterwards; all loads are unconditional (clamped address +
  // select) so they stay in flight together.  Thread t owns chunk slots t + i*NT.
  constexpr int NT = NW * 64;
  constexpr int KCH = (KT * (DKP / 8) + NT - 1) / NT;      // 16-B K chunks per thread
  constexpr int VCR = KT / 4;                              // 8-B chunks (4 keys) per V^T row
  constexpr int VCH = (DT * 32 * VCR + NT - 1) / NT;       // 8-B V^T chunks per thread
  uint4 kreg[KCH];
  uint2 vreg[VCH];
  // Raw buffer loads: the per-thread byte offset is fixed (computed once), the tile advances through the scalar
  // offset, and everything that does not exist -- keys >= Tk (past the end of this sample's K rows), chunks beyond D,
  // threads without a chunk -- is an out-of-range offset that the hardware answers with zeros: no address VALU and
  // no predicates in the loop.
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rsK =
      __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, (int)(((long)(Tk - 1) * ldk + D) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV =
      __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, (int)(((long)(D - 1) * ldvt + ldvt) * 2), 0x00020000);
  unsigned k_off[KCH], v_off[VCH];
#pragma unroll
  for (int i = 0; i < KCH; ++i) {
    const int c = tid + i * NT;
    const int key = c / (DKP / 8), ch = c - key * (DKP / 8);
    const bool ok = (c < KT * (DKP / 8)) && (ch * 8 < D);
    k_off[i] = ok ? (unsigned)((key * ldk + ch * 8) * 2) : OOB;
  }
#pragma unroll
  for (int i = 0; i < VCH; ++i) {
    const int c = tid + i * NT;
    const int d = c / VCR, ch = c - d * VCR;
    const bool ok = (c < DT * 32 * VCR) && (d < D);
    v_off[i] = ok ? (unsigned)((d * ldvt + ch * 4) * 2) : OOB;
  }
  auto gfetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsK, (int)k_off[i], k0 * ldk * 2, 0);
      kreg[i] = make_uint4(v[0], v[1], v[2], v[3]);
    }
#pragma unroll
    for (int i = 0; i < VCH; ++i) {
      typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsV, (int)v_off[i], k0 * 2, 0);
      vreg[i] = make_uint2(v[0], v[1]);
    }
    if (k0 + KT > Tk) {   // ragged last tile only (e.g. 33 context tokens): the row padding may hold stale bits
#pragma unroll
      for (int i = 0; i < VCH; ++i) {
        const int ch = (tid + i * NT) % VCR;
        const int nvalid = Tk - (k0 + ch * 4);   // keys of this 4-key chunk that exist
        uint2 v = vreg[i];
        if (nvalid < 4) {
          if (nvalid < 3) v.y = 0; else v.y &= 0xFFFFu;
          if (nvalid < 2) v.x &= 0xFFFFu;
          if (nvalid < 1) v.x = 0;
        }
        vreg[i] = v;
      }
    }
  };
  auto sstore = [&](int buf) {
    bf16_t* k_ = sK + buf * (KT * KROW);
    bf16_t* v_ = sV + buf * (DT * 32 * VROW);
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
      const int c = tid + i * NT;
      const int key = c / (DKP / 8), ch = c - key * (DKP / 8);
      if (c < KT * (DKP / 8)) *reinterpret_cast<uint4*>(&k_[key * KROW + ch * 8]) = kreg[i];
    }
#pragma unroll
    for (int i = 0; i < VCH; ++i) {
      const int c = tid + i * NT;
      const int d = c / VCR, ch = c - d * VCR;
      if (c < DT * 32 * VCR && !(ONES && d == D)) *reinterpret_cast<uint2*>(&v_[d * VROW + ch * 4]) = vreg[i];
    }
  };

  gfetch(0);
  // Long unrolled instantiations (D >= 128: 9-14 KB of code, run on 16 / 64-token maps where the whole kernel is a few k cycles of work):
  // the code behind the pc through the vector path, BEHIND the first operand requests (in front of them it delays the data: measured
  // slower for the D = 40 / 80 kernels, experiments/round5_measured_and_dropped.md section 16).
  DfTouch code_touch = DfTouch();
  if constexpr (D >= 128) code_touch = df_entry_touch(0);
  sstore(0);
  if (ONES) {                       // row D of both V^T buffers = 1.0 (never overwritten: sstore skips it)
#if defined(DF_OPERAND_F16)
    constexpr uint32_t ONE2 = 0x3C003C00u;
#else
    constexpr uint32_t ONE2 = 0x3F803F80u;
#endif
    for (int i = tid; i < 2 * (KT / 2); i += NT) {
      const int b = i / (KT / 2), k2 = i - b * (KT / 2);
      *reinterpret_cast<uint32_t*>(&sV[b * (DT * 32 * VROW) + D * VROW + 2 * k2]) = ONE2;
    }
  }
  __syncthreads();

  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = kt * KT;
    const int buf = kt & 1;
    const bf16_t* cK = sK + buf * (KT * KROW);
    const bf16_t* cV = sV + buf * (DT * 32 * VROW);
    if (kt + 1 < ntiles) gfetch(k0 + KT);

    // ---- S^T = K Q^T, KS independent 32-key sub-tiles
    f32x16 s[KS];
#pragma unroll
    for (int u = 0; u < KS; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[u][r] = 0.f;
#pragma unroll
    for (int c = 0; c < DKC; ++c)
#pragma unroll
      for (int u = 0; u < KS; ++u) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&cK[(u * 32 + l31) * KROW + 16 * c + 8 * lh]);
        s[u] = DF_MFMA_32x32x16(kf, qf[c], s[u]);
      }
    // ---- online softmax over this lane's 16 KS keys (+ partner half-wave)
    // The softmax scale (x log2 e) is applied inside the exponent's FMA: the maximum is taken over the RAW scores (the scale
    // is positive) and scaled once, p = exp2(fma(s, scale, -m)) -- 16 multiplies per tile fewer than scaling every score.
    if (k0 + KT > Tk) {          // ragged last tile only: mask keys that do not exist
#pragma unroll
      for (int u = 0; u < KS; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + u * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          s[u][r] = (key < Tk) ? s[u][r] : -INFINITY;
        }
    }
    float mxu[KS];
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      mxu[u] = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) mxu[u] = fmaxf(mxu[u], s[u][r]);
    }
    float mx = mxu[0];
#pragma unroll
    for (int u = 1; u < KS; ++u) mx = fmaxf(mx, mxu[u]);
    mx = half_max(mx) * scale_log2e;              // the partner half-wave holds the other 16 keys of this query
    const float m_new = fmaxf(m_run, mx);          // finite: every tile has >= 1 valid key
    float psu[KS];
    uint32_t pk[KS][8];
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      psu[u] = 0.f;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[u][r], scale_log2e, -m_new)),
                    p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[u][r + 1], scale_log2e, -m_new));
        if (!ONES) psu[u] += p0 + p1;
        pk[u][r >> 1] = pack_bf2_bounded(p0, p1);      // p in [0, 1]
      }
    }
    float ps = psu[0];
#pragma unroll
    for (int u = 1; u < KS; ++u) ps += psu[u];
    if (__any(m_new != m_run)) {                   // wave-uniform: rescale only when some row's running max moved
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);      // first tile: exp2(-inf) = 0
      if (!ONES) l_run *= alpha;
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
      m_run = m_new;
    }
    if (!ONES) l_run += ps;
    // ---- O^T += V^T P^T
#pragma unroll
    for (int u = 0; u < KS; ++u)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        uint4 pv = make_uint4(pk[u][4 * c], pk[u][4 * c + 1], pk[u][4 * c + 2], pk[u][4 * c + 3]);
        const bf16x8 pf = *reinterpret_cast<bf16x8*>(&pv);
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          const bf16_t* vr = &cV[(t * 32 + l31) * VROW + u * 32 + 16 * c + 4 * lh];
          const uint2 v0 = *reinterpret_cast<const uint2*>(vr);
          const uint2 v1 = *reinterpret_cast<const uint2*>(vr + 8);
          uint4 vv = make_uint4(v0.x, v0.y, v1.x, v1.y);
          const bf16x8 vf = *reinterpret_cast<bf16x8*>(&vv);
          o[t] = DF_MFMA_32x32x16(vf, pf, o[t]);
        }
      }
    if (kt + 1 < ntiles) sstore(buf ^ 1);   // the other buffer was last read in iteration kt-1
    __syncthreads();
  }

  // ---- normalise and store: lane (q, lh) holds O[q][32t + (r&3) + 8(r>>2) + 4lh]
  float l_tot;
  if (ONES) {
    const float mine = o[ONE_T][ONE_R];          // valid in half-wave 0
    const float other = __shfl_xor(mine, 32);
    l_tot = lh == 0 ? mine : other;
  } else {
    l_tot = l_run + __shfl_xor(l_run, 32);
  }
  const float inv = 1.0f / l_tot;
  if (qv) {
    bf16_t* orow = O + ((long)n * Tq + q) * ldo + h * D;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = 32 * t + 8 * g + 4 * lh;
        if (d0 < D) {   // D % 4 == 0
          uint2 w;
          w.x = pack_bf2(o[t][4 * g] * inv, o[t][4 * g + 1] * inv);
          w.y = pack_bf2(o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
          *reinterpret_cast<uint2*>(orow + d0) = w;
        }
      }
  }
  if constexpr (D >= 128) df_entry_touch_end(code_touch);
}

template <int D>
hipError_t launch_d(const uint16_t* Q, int ldq, const uint16_t* K, int ldk, const uint16_t* Vt, int ldvt,
                    uint16_t* O, int ldo, int N, int heads, int Tq, int Tk, float scale, hipStream_t s) {
  const float sl2 = scale * 1.4426950408889634f;
  if constexpr (D <= 80) {        // two 64-key buffers of K and V^T fit beside each other (<= 35 KB)
    constexpr bool no_ks2 = false;
    if (Tq >= 128 && Tk % 64 == 0 && !no_ks2) {
      dim3 grid((Tq + 127) / 128, heads, N);
      hipLaunchKernelGGL((attention_kernel<D, 4, 2>), grid, dim3(256), 0, s, Q, ldq, K, ldk, Vt, ldvt, heads, Tq, Tk, sl2, O, ldo);
      return hipGetLastError();
    }
  }
  if (Tq >= 128) {
    dim3 grid((Tq + 127) / 128, heads, N);
    hipLaunchKernelGGL((attention_kernel<D, 4>), grid, dim3(256), 0, s, Q, ldq, K, ldk, Vt, ldvt, heads, Tq, Tk, sl2,
                       O, ldo);
  } else if (Tq >= 64) {
    dim3 grid((Tq + 63) / 64, heads, N);
    hipLaunchKernelGGL((attention_kernel<D, 2>), grid, dim3(128), 0, s, Q, ldq, K, ldk, Vt, ldvt, heads, Tq, Tk, sl2,
                       O, ldo);
  } else {
    dim3 grid((Tq + 31) / 32, heads, N);
    hipLaunchKernelGGL((attention_kernel<D, 1>), grid, dim3(64), 0, s, Q, ldq, K, ldk, Vt, ldvt, heads, Tq, Tk, sl2,
                       O, ldo);
  }
  return hipGetLastError();
}

}  // namespace

// Head dims: 40 / 80 / 160 are the Stage-2 UNet's (320 / 640 / 1280 channels over 8 heads), 32 / 64 / 128 the classifier's and the
// power-of-two configurations'; 16 / 24 / 48 / 56 / 72 / 96 / 112 / 192 cover the other model_channels x channel_mult / num_heads
// quotients a UNetModel constructor admits (192 channels over 2 / 4 / 8 heads, 384 over 4 / 8, 448 over 8, 576 over 8 ...).
bool attention_supported(int D) {
  switch (D) {
    case 16: case 24: case 32: case 40: case 48: case 56: case 64: case 72: case 80: case 96: case 112: case 128: case 160: case 192:
      return true;
    default: return false;
  }
}

hipError_t launch_attention(const uint16_t* Q, int ldq, const uint16_t* K, int ldk, const uint16_t* Vt, int ldvt,
                            uint16_t* O, int ldo, int N, int heads, int D, int Tq, int Tk, float scale,
                            hipStream_t s) {
  if (ldvt % 32 != 0 || ldvt < ((Tk + 31) / 32) * 32) return hipErrorInvalidValue;
  switch (D) {
    case 16:  return launch_d<16>(Q, ldq, K, ldk, Vt, ldvt, O, ldo, N, heads, Tq, Tk, scale, s);
    case 24:  return launch_d<24>(Q, ldq, K, ldk, Vt, ldvt, O, ldo, N, heads, Tq, Tk, scale, s);
    case 48:  return launch_d<48>(Q, ldq, K, ldk, Vt, ldvt, O, ldo, N, heads, Tq, Tk, scale, s);
    case 56:  return launch_d<56>(Q, ldq, K, ldk, Vt, ldvt, O, ldo, N, heads, Tq, Tk, scale, s);
    case 72:  return launch_d<72>(Q, ldq, K, ldk, Vt, ldvt, O, ldo, N, heads, Tq, Tk, scale, s);
    case 96:  return launch_d<96>(Q, ldq, K, ldk, Vt, ldvt, O, ldo, N, heads, Tq, Tk, scale, s);
    case 112: return launch_d<112>(Q, ldq, K, ldk, Vt, ldvt, O, ldo, N, heads, Tq, Tk, scale, s);
    case 192: return launch_d<192>(Q, ldq, K, ldk, Vt, ldvt, O, ldo, N, heads, Tq, Tk, scale, s);
    case 32:  return launch_d<32>(Q, ldq, K, ldk, Vt, ldvt, O, ldo, N, heads, Tq, Tk, scale, s);
    case 40:  return launch_d<40>(Q, ldq, K, ldk, Vt, ldvt, O, ldo, N, heads, Tq, Tk, scale, s);
    case 64:  return launch_d<64>(Q, ldq, K, ldk, Vt, ldvt, O, ldo, N, heads, Tq, Tk, scale, s);
    case 80:  return launch_d<80>(Q, ldq, K, ldk, Vt, ldvt, O, ldo, N, heads, Tq, Tk, scale, s);
    case 128: return launch_d<128>(Q, ldq, K, ldk, Vt, ldvt, O, ldo, N, heads, Tq, Tk, scale, s);
    case 160: return launch_d<160>(Q, ldq, K, ldk, Vt, ldvt, O, ldo, N, heads, Tq, Tk, scale, s);
    default:  return hipErrorInvalidValue;
  }
}
