// Input-gradient (VJP) kernels for the alignment classifier (double guidance needs d log p / d x,
// ddim.py:333-341 / dpm_solver.py:1340-1349).  Only gradients w.r.t. activations are computed; the backward
// data GEMMs (conv^T, Linear^T) reuse gemm_bf16_kernel with transposed / flipped weight packings.
// Gradients travel as fp32 NHWC; every kernel that feeds a GEMM also emits the bf16 copy the GEMM consumes.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

namespace {

// ------------------------------------------------------------------ GroupNorm(32) [+SiLU] backward
// y = act(xhat * gamma + beta), xhat = (x - mean) * rstd over the (HW x cpg) slab of one (sample, group).
//   g_i   = dy_i * act'(.) * gamma_c
//   dx_i  = rstd * (g_i - mean(g) - xhat_i * mean(g * xhat))       [+ addend_i]
// One block per (group, sample); statistics are recomputed from the saved fp32 input.
__global__ void groupnorm_bwd_kernel(const float* __restrict__ x, int ld, int HW, int C, int cpg,
                                     const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                     int silu, const float* __restrict__ dy, int lddy, const float* __restrict__ addend,
                                     int ldadd, float* __restrict__ dx, int lddx, bf16_t* __restrict__ dx_b16) {
  __shared__ float red[16];
  const int g = (blockIdx.x & 7) * 4 + (blockIdx.x >> 3), n = blockIdx.y;
  const long rowbase = (long)n * HW;
  const int c0 = g * cpg, items = HW * cpg;
  const float cnt = (float)items;
  float s = 0.f;
  for (int i = threadIdx.x; i < items; i += blockDim.x) {
    const int px = i / cpg, j = i - px * cpg;
    s += x[(rowbase + px) * ld + c0 + j];
  }
  const float mean = block_sum(s, red) / cnt;
  float q = 0.f;
  for (int i = threadIdx.x; i < items; i += blockDim.x) {
    const int px = i / cpg, j = i - px * cpg;
    const float d = x[(rowbase + px) * ld + c0 + j] - mean;
    q += d * d;
  }
  const float rstd = rsqrtf(block_sum(q, red) / cnt + eps);
  float sg = 0.f, sgx = 0.f;
  for (int i = threadIdx.x; i < items; i += blockDim.x) {
    const int px = i / cpg, j = i - px * cpg, c = c0 + j;
    const float xh = (x[(rowbase + px) * ld + c] - mean) * rstd;
    float gi = dy[(rowbase + px) * lddy + c];
    if (silu) {
      const float u = xh * gamma[c] + beta[c];
      const float sig = 1.0f / (1.0f + __expf(-u));
      gi *= sig * (1.0f + u * (1.0f - sig));
    }
    gi *= gamma[c];
    sg += gi;
    sgx += gi * xh;
  }
  const float mg = block_sum(sg, red) / cnt;
  const float mgx = block_sum(sgx, red) / cnt;
  for (int i = threadIdx.x; i < items; i += blockDim.x) {
    const int px = i / cpg, j = i - px * cpg, c = c0 + j;
    const float xh = (x[(rowbase + px) * ld + c] - mean) * rstd;
    float gi = dy[(rowbase + px) * lddy + c];
    if (silu) {
      const float u = xh * gamma[c] + beta[c];
      const float sig = 1.0f / (1.0f + __expf(-u));
      gi *= sig * (1.0f + u * (1.0f - sig));
    }
    gi *= gamma[c];
    float r = rstd * (gi - mg - xh * mgx);
    if (addend) r += addend[(rowbase + px) * ldadd + c];
    dx[(rowbase + px) * lddx + c] = r;
    if (dx_b16) dx_b16[(rowbase + px) * C + c] = f2bf(r);
  }
}

// ------------------------------------------------------------------ LayerNorm backward (one wave per row)
//   g = dy * gamma ; dx = rstd * (g - mean(g) - xhat * mean(g * xhat))  [+ addend]
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, int rows, int C,
                                                            const float* __restrict__ gamma, float eps,
                                                            const float* __restrict__ dy,
                                                            const float* __restrict__ addend,
                                                            float* __restrict__ dx, bf16_t* __restrict__ dx_b16) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (long)row * C;
  const float* dr = dy + (long)row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c];
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float d = xr[c] - mean;
    q += d * d;
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  float sg = 0.f, sgx = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float gi = dr[c] * gamma[c], xh = (xr[c] - mean) * rstd;
    sg += gi;
    sgx += gi * xh;
  }
  const float mg = wave_sum(sg) / (float)C, mgx = wave_sum(sgx) / (float)C;
  for (int c = lane; c < C; c += 64) {
    const float gi = dr[c] * gamma[c], xh = (xr[c] - mean) * rstd;
    float r = rstd * (gi - mg - xh * mgx);
    if (addend) r += addend[(long)row * C + c];
    dx[(long)row * C + c] = r;
    if (dx_b16) dx_b16[(long)row * C + c] = f2bf(r);
  }
}

// ------------------------------------------------------------------ GEGLU forward (unfused) / backward
// u = [x | gate] bf16 [rows][2H] (bias already added); y = x * gelu(gate)
__global__ void geglu_fwd_kernel(const bf16_t* __restrict__ u, bf16_t* __restrict__ y, long rows, int H) {
  const long total = rows * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / H;
    const int c = (int)(i - r * H);
    const float xv = bf2f(u[r * 2 * H + c]), gv = bf2f(u[r * 2 * H + H + c]);
    y[i] = f2bf(xv * gelu_erf(gv));
  }
}
// du = [dy * gelu(g) | dy * x * gelu'(g)],  gelu'(g) = Phi(g) + g * phi(g)
__global__ void geglu_bwd_kernel(const bf16_t* __restrict__ u, const float* __restrict__ dy,
                                 bf16_t* __restrict__ du, long rows, int H) {
  const long total = rows * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / H;
    const int c = (int)(i - r * H);
    const float xv = bf2f(u[r * 2 * H + c]), gv = bf2f(u[r * 2 * H + H + c]);
    const float d = dy[i];
    const float Phi = 0.5f * (1.0f + erf_as(gv * 0.70710678118654752440f));
    const float phi = 0.39894228040143267794f * __expf(-0.5f * gv * gv);
    du[r * 2 * H + c] = f2bf(d * gv * Phi);
    du[r * 2 * H + H + c] = f2bf(d * xv * (Phi + gv * phi));
  }
}

// ------------------------------------------------------------------ attention backward (small T, VALU)
// One block per (head, sample).  Phase A: thread per query row -> softmax statistics (m, l), delta = sum_j p dp,
// and dQ.  Phase B: thread per key -> dK, dV from the saved row statistics.  K / V / Q / dO are staged in LDS as
// fp32; all loops are plain FMAs (the classifier's attention is ~0.1 % of its FLOPs).
template <int D>
__global__ __launch_bounds__(256) void attention_bwd_kernel(const bf16_t* __restrict__ Q, int ldq,
                                                            const bf16_t* __restrict__ K, int ldk,
                                                            const bf16_t* __restrict__ Vt, int ldvt,
                                                            const float* __restrict__ dO, int lddo,
                                                            bf16_t* __restrict__ dQ, int lddq, bf16_t* __restrict__ dK,
                                                            int lddk, bf16_t* __restrict__ dV, int lddv, int heads,
                                                            int Tq, int Tk, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sK = sm;                    // [Tk][D+1]
  float* sV = sK + Tk * (D + 1);     // [Tk][D+1]
  float* sQ = sV + Tk * (D + 1);     // [Tq][D+1]
  float* sD = sQ + Tq * (D + 1);     // [Tq][D+1]  dO
  float* sM = sD + Tq * (D + 1);     // [Tq] row max
  float* sL = sM + Tq;               // [Tq] row sum
  float* sDel = sL + Tq;             // [Tq] delta
  const int h = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  for (int i = tid; i < Tk * D; i += 256) {
    const int j = i / D, d = i - j * D;
    sK[j * (D + 1) + d] = bf2f(K[((long)n * Tk + j) * ldk + h * D + d]);
    sV[j * (D + 1) + d] = bf2f(Vt[((long)n * heads + h) * D * ldvt + (long)d * ldvt + j]);
  }
  for (int i = tid; i < Tq * D; i += 256) {
    const int q = i / D, d = i - q * D;
    sQ[q * (D + 1) + d] = bf2f(Q[((long)n * Tq + q) * ldq + h * D + d]);
    sD[q * (D + 1) + d] = dO[((long)n * Tq + q) * lddo + h * D + d];
  }
  __syncthreads();
  // ---- phase A
  for (int q = tid; q < Tq; q += 256) {
    float qv[D], dv[D], dq[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      qv[d] = sQ[q * (D + 1) + d];
      dv[d] = sD[q * (D + 1) + d];
      dq[d] = 0.f;
    }
    float m = -INFINITY;
    for (int j = 0; j < Tk; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) s += qv[d] * sK[j * (D + 1) + d];
      m = fmaxf(m, s * scale);
    }
    float l = 0.f, del = 0.f;
    for (int j = 0; j < Tk; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        s += qv[d] * sK[j * (D + 1) + d];
        dp += dv[d] * sV[j * (D + 1) + d];
      }
      const float e = __expf(s * scale - m);
      l += e;
      del += e * dp;
    }
    del /= l;
    for (int j = 0; j < Tk; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        s += qv[d] * sK[j * (D + 1) + d];
        dp += dv[d] * sV[j * (D + 1) + d];
      }
      const float pj = __expf(s * scale - m) / l;
      const float ds = pj * (dp - del) * scale;
#pragma unroll
      for (int d = 0; d < D; ++d) dq[d] += ds * sK[j * (D + 1) + d];
    }
    sM[q] = m;
    sL[q] = l;
    sDel[q] = del;
#pragma unroll
    for (int d = 0; d < D; ++d) dQ[((long)n * Tq + q) * lddq + h * D + d] = f2bf(dq[d]);
  }
  if (!dK) return;
  __syncthreads();
  // ---- phase B
  for (int j = tid; j < Tk; j += 256) {
    float kv[D], vv[D], dk[D], dvv[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      kv[d] = sK[j * (D + 1) + d];
      vv[d] = sV[j * (D + 1) + d];
      dk[d] = 0.f;
      dvv[d] = 0.f;
    }
    for (int q = 0; q < Tq; ++q) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        s += sQ[q * (D + 1) + d] * kv[d];
        dp += sD[q * (D + 1) + d] * vv[d];
      }
      const float pj = __expf(s * scale - sM[q]) / sL[q];
      const float ds = pj * (dp - sDel[q]) * scale;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        dk[d] += ds * sQ[q * (D + 1) + d];
        dvv[d] += pj * sD[q * (D + 1) + d];
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
      dK[((long)n * Tk + j) * lddk + h * D + d] = f2bf(dk[d]);
      dV[((long)n * Tk + j) * lddv + h * D + d] = f2bf(dvv[d]);
    }
  }
}

// ------------------------------------------------------------------ attention backward, tiled (any T; VALU)
// The two kernels above keep a whole (head, sample) in LDS -- the shapes of the reference's classifier on the 16 x 64 latent
// (<= 256 tokens per map).  Longer maps (attention at the full resolution, a wider latent through `size_len`) take this pair: one
// wavefront per 64 query rows (thread = row: statistics, delta and dQ as in phase A above, K / V streamed through LDS in tiles
// of 64 keys, three passes), then one wavefront per 64 keys (thread = key: dK, dV as in phase B, Q / dO and the saved row
// statistics streamed in tiles of 64 queries).  Same arithmetic, same summation order per row / key as the resident kernel.
template <int D>
__global__ __launch_bounds__(64) void attention_bwd_tiled_dq_kernel(const bf16_t* __restrict__ Q, int ldq,
                                                                    const bf16_t* __restrict__ K, int ldk,
                                                                    const bf16_t* __restrict__ Vt, int ldvt,
                                                                    const float* __restrict__ dO, int lddo,
                                                                    bf16_t* __restrict__ dQ, int lddq, float* __restrict__ stats,
                                                                    int heads, int Tq, int Tk, float scale) {
  __shared__ float sK[64 * (D + 1)], sV[64 * (D + 1)];
  const int tid = threadIdx.x, q = blockIdx.x * 64 + tid, h = blockIdx.y, n = blockIdx.z;
  const bool valid = q < Tq;
  const int qc = valid ? q : Tq - 1;
  float qv[D], dv[D], dq[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    qv[d] = bf2f(Q[((long)n * Tq + qc) * ldq + h * D + d]);
    dv[d] = dO[((long)n * Tq + qc) * lddo + h * D + d];
    dq[d] = 0.f;
  }
  auto stage = [&](int k0, bool with_v) {
    __syncthreads();
    for (int i = tid; i < 64 * D; i += 64) {
      const int j = i / D, d = i - j * D;
      sK[j * (D + 1) + d] = (k0 + j < Tk) ? bf2f(K[((long)n * Tk + k0 + j) * ldk + h * D + d]) : 0.f;
    }
    if (with_v)
      for (int i = tid; i < 64 * D; i += 64) {
        const int d = i >> 6, j = i & 63;
        sV[j * (D + 1) + d] = (k0 + j < Tk) ? bf2f(Vt[((long)n * heads + h) * D * ldvt + (long)d * ldvt + k0 + j]) : 0.f;
      }
    __syncthreads();
  };
  float m = -INFINITY;
  for (int k0 = 0; k0 < Tk; k0 += 64) {
    stage(k0, false);
    const int nj = min(64, Tk - k0);
    for (int j = 0; j < nj; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) s += qv[d] * sK[j * (D + 1) + d];
      m = fmaxf(m, s * scale);
    }
  }
  float l = 0.f, del = 0.f;
  for (int k0 = 0; k0 < Tk; k0 += 64) {
    stage(k0, true);
    const int nj = min(64, Tk - k0);
    for (int j = 0; j < nj; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        s += qv[d] * sK[j * (D + 1) + d];
        dp += dv[d] * sV[j * (D + 1) + d];
      }
      const float e = __expf(s * scale - m);
      l += e;
      del += e * dp;
    }
  }
  del /= l;
  for (int k0 = 0; k0 < Tk; k0 += 64) {
    stage(k0, true);
    const int nj = min(64, Tk - k0);
    for (int j = 0; j < nj; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        s += qv[d] * sK[j * (D + 1) + d];
        dp += dv[d] * sV[j * (D + 1) + d];
      }
      const float pj = __expf(s * scale - m) / l;
      const float ds = pj * (dp - del) * scale;
#pragma unroll
      for (int d = 0; d < D; ++d) dq[d] += ds * sK[j * (D + 1) + d];
    }
  }
  if (!valid) return;
  if (stats) {
    float* st = stats + (((long)n * heads + h) * Tq + q) * 3;
    st[0] = m;
    st[1] = l;
    st[2] = del;
  }
#pragma unroll
  for (int d = 0; d < D; ++d) dQ[((long)n * Tq + q) * lddq + h * D + d] = f2bf(dq[d]);
}

template <int D>
__global__ __launch_bounds__(64) void attention_bwd_tiled_dkv_kernel(const bf16_t* __restrict__ Q, int ldq,
                                                                     const bf16_t* __restrict__ K, int ldk,
                                                                     const bf16_t* __restrict__ Vt, int ldvt,
                                                                     const float* __restrict__ dO, int lddo,
                                                                     bf16_t* __restrict__ dK, int lddk, bf16_t* __restrict__ dV,
                                                                     int lddv, const float* __restrict__ stats, int heads, int Tq,
                                                                     int Tk, float scale) {
  __shared__ float sQ[64 * (D + 1)], sD[64 * (D + 1)], sS[64 * 3];
  const int tid = threadIdx.x, j = blockIdx.x * 64 + tid, h = blockIdx.y, n = blockIdx.z;
  const bool valid = j < Tk;
  const int jc = valid ? j : Tk - 1;
  float kv[D], vv[D], dk[D], dvv[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    kv[d] = bf2f(K[((long)n * Tk + jc) * ldk + h * D + d]);
    vv[d] = bf2f(Vt[((long)n * heads + h) * D * ldvt + (long)d * ldvt + jc]);
    dk[d] = 0.f;
    dvv[d] = 0.f;
  }
  for (int q0 = 0; q0 < Tq; q0 += 64) {
    __syncthreads();
    for (int i = tid; i < 64 * D; i += 64) {
      const int q = i / D, d = i - q * D;
      const bool in = q0 + q < Tq;
      sQ[q * (D + 1) + d] = in ? bf2f(Q[((long)n * Tq + q0 + q) * ldq + h * D + d]) : 0.f;
      sD[q * (D + 1) + d] = in ? dO[((long)n * Tq + q0 + q) * lddo + h * D + d] : 0.f;
    }
    for (int i = tid; i < 64 * 3; i += 64)
      sS[i] = (q0 + i / 3 < Tq) ? stats[(((long)n * heads + h) * Tq + q0) * 3 + i] : 1.f;
    __syncthreads();
    const int nq = min(64, Tq - q0);
    for (int q = 0; q < nq; ++q) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        s += sQ[q * (D + 1) + d] * kv[d];
        dp += sD[q * (D + 1) + d] * vv[d];
      }
      const float pj = __expf(s * scale - sS[q * 3]) / sS[q * 3 + 1];
      const float ds = pj * (dp - sS[q * 3 + 2]) * scale;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        dk[d] += ds * sQ[q * (D + 1) + d];
        dvv[d] += pj * sD[q * (D + 1) + d];
      }
    }
  }
  if (!valid) return;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    dK[((long)n * Tk + j) * lddk + h * D + d] = f2bf(dk[d]);
    dV[((long)n * Tk + j) * lddv + h * D + d] = f2bf(dvv[d]);
  }
}

// ------------------------------------------------------------------ attention backward, MFMA (D = 32, Tk <= 256)
// One block per (head, sample); wavefront w owns the 32-key tile w (NKT = ceil(Tk/32) <= 8 wavefronts) and keeps
// dK^T / dV^T of its keys in MFMA accumulators while the block walks the query tiles.  Per (query tile, key tile):
//   S^T = K Q^T, dP^T = V dO^T   (keys x queries; lane = query: softmax statistics are lane-local + one shfl)
//   S   = Q K^T, dP   = dO V^T   (queries x keys; lane = key: the orientation whose C layout is a valid B operand
//                                 for contractions over QUERIES -- recomputing beats transposing, D is only 32)
//   row max / sum / delta = sum_j p dp are reduced over the key tiles through LDS in wavefront order (deterministic)
//   dV^T += dO^T P,  dK^T += Q^T dS   (contraction over the 32 queries, in the row order of the C layout)
//   dQ^T  = sum_w K_w^T dS_w^T        (contraction over keys; per-wavefront partials summed in LDS in fixed order)
// P and dS are rounded to the operand type before the second MFMAs, exactly like the forward kernel.
// LDS: Q, K, V, dO row-major [T][40] in operand type; "transposed" A fragments are gathered with 2-byte reads.
__global__ __launch_bounds__(512) void attention_bwd_mfma_kernel(const bf16_t* __restrict__ Q, int ldq,
                                                                 const bf16_t* __restrict__ K, int ldk,
                                                                 const bf16_t* __restrict__ Vt, int ldvt,
                                                                 const float* __restrict__ dO, int lddo,
                                                                 bf16_t* __restrict__ dQ, int lddq,
                                                                 bf16_t* __restrict__ dK, int lddk,
                                                                 bf16_t* __restrict__ dV, int lddv, int heads, int Tq,
                                                                 int Tk, float scale) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int D = 32, DP = 40;
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  const int NKT = (Tk + 31) >> 5, NQT = (Tq + 31) >> 5;
  const int TkP = NKT * 32, TqP = NQT * 32;
  bf16_t* sK = reinterpret_cast<bf16_t*>(smraw);           // [TkP][DP]
  bf16_t* sV = sK + TkP * DP;                              // [TkP][DP]
  bf16_t* sQ = sV + TkP * DP;                              // [TqP][DP]
  bf16_t* sD = sQ + TqP * DP;                              // [TqP][DP]  dO
  float* red = reinterpret_cast<float*>(sD + TqP * DP);    // [3][8][32]  per-wave max / sum / sum(e*dp)
  float* stat = red + 3 * 8 * 32;                          // [3][32]     row max, 1/sum, delta of the query tile
  float* part = stat + 3 * 32;                             // [8][32][33] dQ^T partials (d, q)
  const int h = blockIdx.x, n = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const float c2 = scale * 1.44269504088896340736f;

  // ---- stage (zero padded to whole tiles)
  for (int i = tid; i < TkP * D; i += nthr) {
    const int j = i >> 5, d = i & 31;
    bf16_t kv = 0, vv = 0;
    if (j < Tk) {
      kv = K[((long)n * Tk + j) * ldk + h * D + d];
      vv = Vt[((long)n * heads + h) * D * ldvt + (long)d * ldvt + j];
    }
    sK[j * DP + d] = kv;
    sV[j * DP + d] = vv;
  }
  for (int i = tid; i < TqP * D; i += nthr) {
    const int q = i >> 5, d = i & 31;
    bf16_t qv = 0, dv = 0;
    if (q < Tq) {
      qv = Q[((long)n * Tq + q) * ldq + h * D + d];
      dv = f2bf(dO[((long)n * Tq + q) * lddo + h * D + d]);
    }
    sQ[q * DP + d] = qv;
    sD[q * DP + d] = dv;
  }
  __syncthreads();

  const int k0 = wid * 32;                                 // this wavefront's key tile
  // row-major fragments (A or B operand: row = l31, 8 consecutive d at 16*s + 8*lh)
  auto frag = [&](const bf16_t* base, int row0, int s2) {
    return *reinterpret_cast<const bf16x8*>(&base[(row0 + l31) * DP + 16 * s2 + 8 * lh]);
  };
  // "transposed" A fragment: row m = d (l31), contraction index = tile rows in C-layout order of k-step s2:
  // rows {0..3, 8..11} + 4*lh + 16*s2
  auto fragT = [&](const bf16_t* base, int row0, int s2) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ra = row0 + 16 * s2 + 4 * lh + ((2 * i) & 3) + 8 * ((2 * i) >> 2);
      const uint32_t lo = base[ra * DP + l31], hi = base[(ra + 1) * DP + l31];
      w[i] = lo | (hi << 16);
    }
    uint4 v = make_uint4(w[0], w[1], w[2], w[3]);
    return *reinterpret_cast<bf16x8*>(&v);
  };
  const bf16x8 kf0 = frag(sK, k0, 0), kf1 = frag(sK, k0, 1), vf0 = frag(sV, k0, 0), vf1 = frag(sV, k0, 1);
  const bf16x8 kt0 = fragT(sK, k0, 0), kt1 = fragT(sK, k0, 1);

  f32x16 dkT, dvT;                                         // (d x keys): lane key = l31, rows d
#pragma unroll
  for (int r = 0; r < 16; ++r) dkT[r] = dvT[r] = 0.f;
  const f32x16 zero16 = dkT;

  for (int qt = 0; qt < NQT; ++qt) {
    const int q0 = qt * 32;
    const bf16x8 qf0 = frag(sQ, q0, 0), qf1 = frag(sQ, q0, 1), df0 = frag(sD, q0, 0), df1 = frag(sD, q0, 1);
    // keys x queries
    f32x16 st = DF_MFMA_32x32x16(kf0, qf0, zero16);
    st = DF_MFMA_32x32x16(kf1, qf1, st);
    f32x16 dpt = DF_MFMA_32x32x16(vf0, df0, zero16);
    dpt = DF_MFMA_32x32x16(vf1, df1, dpt);
    // queries x keys
    f32x16 ss = DF_MFMA_32x32x16(qf0, kf0, zero16);
    ss = DF_MFMA_32x32x16(qf1, kf1, ss);
    f32x16 dps = DF_MFMA_32x32x16(df0, vf0, zero16);
    dps = DF_MFMA_32x32x16(df1, vf1, dps);

    // ---- statistics of query l31 over this wavefront's keys (S^T orientation), then over all key tiles
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      st[r] = (key < Tk) ? st[r] * c2 : -INFINITY;
      mx = fmaxf(mx, st[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (lh == 0) red[wid * 32 + l31] = mx;
    __syncthreads();
    float m = red[l31];
    for (int w = 1; w < NKT; ++w) m = fmaxf(m, red[w * 32 + l31]);
    float ls = 0.f, le = 0.f;
    float et[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      et[r] = __builtin_amdgcn_exp2f(st[r] - m);           // masked keys: exp2(-inf) = 0
      ls += et[r];
      le += et[r] * dpt[r];
    }
    ls += __shfl_xor(ls, 32);
    le += __shfl_xor(le, 32);
    if (lh == 0) {
      red[256 + wid * 32 + l31] = ls;
      red[512 + wid * 32 + l31] = le;
    }
    __syncthreads();
    float lsum = 0.f, esum = 0.f;
    for (int w = 0; w < NKT; ++w) {
      lsum += red[256 + w * 32 + l31];
      esum += red[512 + w * 32 + l31];
    }
    const bool qv = (q0 + l31) < Tq;
    const float inv = qv ? 1.0f / lsum : 0.f;              // padded queries contribute nothing
    const float del = esum * inv;
    if (wid == 0 && lh == 0) {
      stat[l31] = m;
      stat[32 + l31] = inv;
      stat[64 + l31] = del;
    }
    // dS^T as B operand (contraction over this tile's keys, C-layout row order)
    uint32_t dst[8];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const float a = et[r] * inv * (dpt[r] - del) * scale, b = et[r + 1] * inv * (dpt[r + 1] - del) * scale;
      dst[r >> 1] = pack_bf2(a, b);
    }
    __syncthreads();                                        // stat visible; red free for the next tile
    // ---- P and dS in the (queries x keys) orientation: lane = key, rows = queries
    uint32_t pp[8], dsp[8];
    const bool kv = (k0 + l31) < Tk;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      float pv[2], dv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int qi = ((r + u) & 3) + 8 * ((r + u) >> 2) + 4 * lh;
        const float e = kv ? __builtin_amdgcn_exp2f(ss[r + u] * c2 - stat[qi]) * stat[32 + qi] : 0.f;
        pv[u] = e;
        dv[u] = e * (dps[r + u] - stat[64 + qi]) * scale;
      }
      pp[r >> 1] = pack_bf2(pv[0], pv[1]);
      dsp[r >> 1] = pack_bf2(dv[0], dv[1]);
    }
    // ---- dV^T += dO^T P ; dK^T += Q^T dS   (two k-steps of 16 queries)
    if (dK) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        uint4 pb = make_uint4(pp[4 * s2], pp[4 * s2 + 1], pp[4 * s2 + 2], pp[4 * s2 + 3]);
        uint4 db = make_uint4(dsp[4 * s2], dsp[4 * s2 + 1], dsp[4 * s2 + 2], dsp[4 * s2 + 3]);
        dvT = DF_MFMA_32x32x16(fragT(sD, q0, s2), *reinterpret_cast<bf16x8*>(&pb), dvT);
        dkT = DF_MFMA_32x32x16(fragT(sQ, q0, s2), *reinterpret_cast<bf16x8*>(&db), dkT);
      }
    }
    // ---- dQ^T partial of this key tile: K_w^T dS_w^T  -> (d x queries), lane = query
    {
      uint4 b0 = make_uint4(dst[0], dst[1], dst[2], dst[3]), b1 = make_uint4(dst[4], dst[5], dst[6], dst[7]);
      f32x16 dq = DF_MFMA_32x32x16(kt0, *reinterpret_cast<bf16x8*>(&b0), zero16);
      dq = DF_MFMA_32x32x16(kt1, *reinterpret_cast<bf16x8*>(&b1), dq);
      float* pw = part + wid * (32 * 33);
#pragma unroll
      for (int r = 0; r < 16; ++r) pw[((r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + l31] = dq[r];
    }
    __syncthreads();
    for (int i = tid; i < 32 * 32; i += nthr) {            // (q, d): sum the key tiles in fixed order
      const int q = i >> 5, d = i & 31;
      float acc = 0.f;
      for (int w = 0; w < NKT; ++w) acc += part[w * (32 * 33) + d * 33 + q];
      if (q0 + q < Tq) dQ[((long)n * Tq + q0 + q) * lddq + h * D + d] = f2bf(acc);
    }
    // (the next iteration's first __syncthreads orders these reads of `part` before its next writes)
  }
  if (dK) {                                                 // lane (key = l31 + k0, lh) holds d = (r&3) + 8(r>>2) + 4lh
    const int key = k0 + l31;
    if (key < Tk) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = 8 * g + 4 * lh;
        uint2 a, b;
        a.x = pack_bf2(dkT[4 * g], dkT[4 * g + 1]); a.y = pack_bf2(dkT[4 * g + 2], dkT[4 * g + 3]);
        b.x = pack_bf2(dvT[4 * g], dvT[4 * g + 1]); b.y = pack_bf2(dvT[4 * g + 2], dvT[4 * g + 3]);
        *reinterpret_cast<uint2*>(&dK[((long)n * Tk + key) * lddk + h * D + d0]) = a;
        *reinterpret_cast<uint2*>(&dV[((long)n * Tk + key) * lddv + h * D + d0]) = b;
      }
    }
  }
#endif
}

// ------------------------------------------------------------------ classifier head backward
// p = sigmoid(z), objective sum log p  ->  dz = 1 - p;  pooled = mean_px(h);  z = w . pooled + b
//   dh[n][px][c] = (1 - p_n) * w[c] / HW          (out_channels == 1)
//   dh_b16: the same values as the next conv's gradient operand, rows of Cp >= C columns, the pad columns zero
__global__ void cls_head_bwd_kernel(const float* __restrict__ prob, const float* __restrict__ w, float* __restrict__ dh,
                                    bf16_t* __restrict__ dh_b16, int N, int HW, int C, int Cp) {
  const long total = (long)N * HW * Cp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cp);
    const long row = i / Cp;
    const int n = (int)(row / HW);
    if (c < C) {
      const float v = (1.0f - prob[n]) * w[c] / (float)HW;
      dh[row * C + c] = v;
      if (dh_b16) dh_b16[i] = f2bf(v);
    } else if (dh_b16) {
      dh_b16[i] = (bf16_t)0;
    }
  }
}

// Linear weight [O][I] fp32 -> transposed bf16 [I][O]
__global__ void pack_linear_t_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int O, int I, int ldo,
                                     int off) {
  const long total = (long)O * I;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int o = (int)(e % O);
    const int i = (int)(e / O);
    out[(long)i * ldo + off + o] = f2bf(w[(long)o * I + i]);
  }
}
// conv OIHW fp32 -> backward-data packing bf16 [I][ky'][kx'][Opad] with (ky', kx') = (2-ky, 2-kx)
__global__ void pack_conv_bwd_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int O, int I, int Opad) {
  const long total = (long)I * 9 * Opad;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int o = (int)(e % Opad);
    long r = e / Opad;
    const int kx = (int)(r % 3);
    r /= 3;
    const int ky = (int)(r % 3);
    const int i = (int)(r / 3);
    out[e] = (o < O) ? f2bf(w[(((long)o * I + i) * 3 + (2 - ky)) * 3 + (2 - kx)]) : (bf16_t)0;
  }
}

inline int grid_for(long n, int block = 256, int cap = 4096) {
  long g = (n + block - 1) / block;
  return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

hipError_t launch_groupnorm_bwd(const float* x, int ld, int N, int HW, int C, const float* gamma, const float* beta,
                                float eps, int silu, const float* dy, int lddy, const float* addend, int ldadd,
                                float* dx, int lddx, uint16_t* dx_b16, hipStream_t s) {
  if (C % 32 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(groupnorm_bwd_kernel, dim3(32, N), dim3(512), 0, s, x, ld, HW, C, C / 32, gamma, beta, eps, silu, dy,
                     lddy, addend, ldadd, dx, lddx, dx_b16);
  return hipGetLastError();
}

hipError_t launch_layernorm_bwd(const float* x, int rows, int C, const float* gamma, float eps, const float* dy,
                                const float* addend, float* dx, uint16_t* dx_b16, hipStream_t s) {
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, rows, C, gamma, eps, dy, addend, dx,
                     dx_b16);
  return hipGetLastError();
}

hipError_t launch_geglu_fwd(const uint16_t* u, uint16_t* y, long rows, int H, hipStream_t s) {
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(grid_for(rows * H)), dim3(256), 0, s, u, y, rows, H);
  return hipGetLastError();
}
hipError_t launch_geglu_bwd(const uint16_t* u, const float* dy, uint16_t* du, long rows, int H, hipStream_t s) {
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid_for(rows * H)), dim3(256), 0, s, u, dy, du, rows, H);
  return hipGetLastError();
}

namespace {
// LDS bytes of the two resident forms; 0 = the form does not take the shape
size_t attn_bwd_mfma_lds(int D, int Tq, int Tk, int lddk, int lddv) {
  if (!(D == 32 && Tk <= 256 && (lddk & 3) == 0 && (lddv & 3) == 0)) return 0;
  const int nkt = (Tk + 31) / 32, nqt = (Tq + 31) / 32;
  const size_t ldm = (size_t)(2 * nkt + 2 * nqt) * 32 * 40 * 2 + (size_t)(3 * 8 * 32 + 3 * 32 + 8 * 32 * 33) * 4;
  return ldm <= 160 * 1024 ? ldm : 0;
}
size_t attn_bwd_valu_lds(int D, int Tq, int Tk) {
  const size_t lds = ((size_t)(2 * Tk + 2 * Tq) * (D + 1) + 3 * (size_t)Tq) * 4;
  return lds <= 160 * 1024 ? lds : 0;
}
}  // namespace

size_t attention_bwd_ws_floats(int N, int heads, int D, int Tq, int Tk, int lddk, int lddv, bool want_dkv) {
  if (attn_bwd_mfma_lds(D, Tq, Tk, lddk, lddv) || attn_bwd_valu_lds(D, Tq, Tk) || !want_dkv) return 0;
  return (size_t)N * heads * Tq * 3;
}

hipError_t launch_attention_bwd(const uint16_t* Q, int ldq, const uint16_t* K, int ldk, const uint16_t* Vt, int ldvt,
                                const float* dO, int lddo, uint16_t* dQ, int lddq, uint16_t* dK, int lddk, uint16_t* dV,
                                int lddv, int N, int heads, int D, int Tq, int Tk, float scale, float* ws, hipStream_t s) {
  if (D != 32 && D != 64) return hipErrorInvalidValue;
  {
    const int nkt = (Tk + 31) / 32;
    const size_t ldm = attn_bwd_mfma_lds(D, Tq, Tk, lddk, lddv);
    if (ldm) {
      static size_t attr_m = 0;
      if (ldm > attr_m) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_bwd_mfma_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldm);
        if (e != hipSuccess) return e;
        attr_m = ldm;
      }
      hipLaunchKernelGGL(attention_bwd_mfma_kernel, dim3(heads, N), dim3(64 * nkt), ldm, s, Q, ldq, K, ldk, Vt, ldvt, dO,
                         lddo, dQ, lddq, dK, lddk, dV, lddv, heads, Tq, Tk, scale);
      return hipGetLastError();
    }
  }
  const size_t lds = attn_bwd_valu_lds(D, Tq, Tk);
  if (!lds) {      // no resident form: the tiled pair (row statistics through `ws`)
    if (dK && !ws) return hipErrorInvalidValue;
    const dim3 gq((Tq + 63) / 64, heads, N), gk((Tk + 63) / 64, heads, N);
#define DF_ABWD_T(DD)                                                                                                             \
  {                                                                                                                               \
    hipLaunchKernelGGL(attention_bwd_tiled_dq_kernel<DD>, gq, dim3(64), 0, s, Q, ldq, K, ldk, Vt, ldvt, dO, lddo, dQ, lddq,         \
                       dK ? ws : nullptr, heads, Tq, Tk, scale);                                                                  \
    if (dK)                                                                                                                       \
      hipLaunchKernelGGL(attention_bwd_tiled_dkv_kernel<DD>, gk, dim3(64), 0, s, Q, ldq, K, ldk, Vt, ldvt, dO, lddo, dK, lddk, dV,   \
                         lddv, ws, heads, Tq, Tk, scale);                                                                         \
  }
    if (D == 32) DF_ABWD_T(32)
    else DF_ABWD_T(64)
#undef DF_ABWD_T
    return hipGetLastError();
  }
#define DF_ABWD(DD)                                                                                                   \
  {                                                                                                                   \
    static size_t attr = 0;                                                                                           \
    if (lds > attr) {                                                                                                 \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_bwd_kernel<DD>),                    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
      if (e != hipSuccess) return e;                                                                                  \
      attr = lds;                                                                                                     \
    }                                                                                                                 \
    hipLaunchKernelGGL(attention_bwd_kernel<DD>, dim3(heads, N), dim3(256), lds, s, Q, ldq, K, ldk, Vt, ldvt, dO, lddo, \
                       dQ, lddq, dK, lddk, dV, lddv, heads, Tq, Tk, scale);                                           \
  }
  if (D == 32) DF_ABWD(32)
  else if (D == 64) DF_ABWD(64)
  else return hipErrorInvalidValue;
#undef DF_ABWD
  return hipGetLastError();
}

hipError_t launch_cls_head_bwd(const float* prob, const float* w, float* dh, uint16_t* dh_b16, int N, int HW, int C, int Cp,
                               hipStream_t s) {
  if (Cp < C) return hipErrorInvalidValue;
  hipLaunchKernelGGL(cls_head_bwd_kernel, dim3(grid_for((long)N * HW * Cp)), dim3(256), 0, s, prob, w, dh, dh_b16, N, HW, C, Cp);
  return hipGetLastError();
}

hipError_t launch_pack_linear_t(const float* w, uint16_t* out, int O, int I, int ldo, int off, hipStream_t s) {
  hipLaunchKernelGGL(pack_linear_t_kernel, dim3(grid_for((long)O * I)), dim3(256), 0, s, w, out, O, I, ldo, off);
  return hipGetLastError();
}
hipError_t launch_pack_conv_bwd(const float* w, uint16_t* out, int O, int I, int Opad, hipStream_t s) {
  hipLaunchKernelGGL(pack_conv_bwd_kernel, dim3(grid_for((long)I * 9 * Opad)), dim3(256), 0, s, w, out, O, I, Opad);
  return hipGetLastError();
}
