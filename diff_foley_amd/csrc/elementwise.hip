// Normalisation, small linears, packing and sampler-update kernels (gfx950).  All HBM/L2-bound.
#include <algorithm>
#include "common.h"
#include "kernels.h"

namespace {

// ------------------------------------------------------------------ GroupNorm(32) [+SiLU] -> bf16
// One block per (group, sample).  Fast path (UNet sizes): the whole group slab (HW x cpg fp32) is held in
// registers -- PER float2 items per thread, loaded unconditionally from clamped addresses so all loads are in
// flight together -- and read from memory exactly once; mean and the centred variance (two-pass, fp32) are
// block reductions.  Large slabs (VAE, up to 1 MB) take the streaming 3-pass path.
// Split-K input (GnSlabs::n > 0): x is the FIRST fp32 partial slab of the producing conv; the kernel sums the n slabs
// (fixed order: deterministic) and adds the conv's bias and per-sample FiLM bias while loading -- the conv's reduce
// launch and the fp32 round trip of its output disappear (the normalised operand is the only consumer of that tensor).
// Split-K PRODUCER of x (GnSlabs::own != null, round 3): the tensor being normalised is itself the not-yet-reduced output of
// the previous GEMM (a ResBlock's conv2 / a SpatialTransformer's merged FF2+proj_out, both followed by the next block's
// GroupNorm).  Channels [0, c_own) of x are summed from the producer's slabs (leading dimension c_own) with its bias and
// fp32 residual added in the reduce kernel's order -- bit-identical to the reduce launch this replaces -- and WRITTEN BACK
// to x (later readers: the residual adds of this block); channels >= c_own (the skip half of a concat buffer) are read from x.
struct GnSlabs {
  int n;
  long stride;                 // floats between consecutive slabs
  const float* bias;           // [C] or null
  const float* rowbias;        // [samples][ld_rowbias] or null
  int ld_rowbias;
  const float* own;            // first slab of x's own producer, or null
  int c_own;
  const float* res;            // fp32 residual of the producer's epilogue, or null
  int ldr;
  float* hout;                 // == x (write-back of the reduced channels)
};
// Thread layout (round 3): thread = (row ty, channel pair tx) with blockDim.x = roundup64(half * R): a thread keeps ONE channel
// pair for all of its PER items (rows ty + k * R), so gamma / beta / bias are loaded once, addresses are affine in k and no
// per-item division or index array exists (the round-2 kernel indexed items linearly: 25 VALU instructions of integer division
// per item and, at PER = 16, scratch spills -- the norm was VALU-bound, its time proportional to the element count).
// Argument order (round 4): what the first loads need -- x, ld, HW, cpg, R and the slab switch (n, own, c_own, stride) -- comes
// first, inside the 16 dwords that are preloaded into SGPRs at wavefront launch (build.sh); everything behind them arrives by
// s_load while the loads are already in flight.  `sl` carries the same four values again; the kernel reads the scalars.
template <int PER>
__global__ void groupnorm_reg_kernel(const float* __restrict__ x, int ld, int HW, int cpg, int R, int sl_n, int nsamp,
                                     const float* __restrict__ sl_own, int sl_c_own, long sl_stride,
                                     bf16_t* __restrict__ out, int ldo, int silu, int C,
                                     const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                     bf16_t* __restrict__ raw, GnSlabs sl) {
  df_fp16_hw_clamp();            // the stores pack with pack_bf2_hw
  __shared__ __attribute__((aligned(16))) float red[16], red2[16];      // one array per reduction: neither needs a barrier in front of its first (only) use
  // block b runs on XCD b%8.  Samples in multiples of 8 (the UNet's CFG batch): ALL groups of a sample on one XCD (b = g * nsamp +
  // n), so every 128-B line of the sample's rows is read into, and written back from, exactly one L2.  Otherwise: the 4
  // neighbouring groups that share lines on one XCD (b = n * 32 + (XCD-major group index)).
  int g, n;
  if ((nsamp & 7) == 0) {
    n = blockIdx.x % nsamp;
    g = blockIdx.x / nsamp;
  } else {
    const int bx = blockIdx.x & 31;
    n = blockIdx.x >> 5;
    g = (bx & 7) * 4 + (bx >> 3);
  }
  const int half = cpg >> 1;
  const int ty = threadIdx.x / half, tx = threadIdx.x - ty * half;
  const bool act = ty < R;                       // the last wavefront may carry idle threads
  const int c = g * cpg + 2 * tx;
  const long row0 = (long)n * HW;
  float2 v[PER];
  // clamped row of item k (loads are unconditional); element offsets fit 32 bits (one tensor / slab < 2^31 elements)
#define DF_ROW(k) min(ty + (k) * R, HW - 1)
  const bool from_slabs = sl_n > 0 && (!sl_own || c < sl_c_own);
  if (from_slabs) {
    // split-K input: the slab loads are issued U slabs at a time (16 independent loads per thread in flight) and added in slab
    // order, so the sum is bit-identical to the reduce kernel's and the latency chain is n / U long, not n
    const int sld = sl_own ? sl_c_own : ld;
    const float* p0 = (sl_own ? sl_own : x) + row0 * sld + c;
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k] = *reinterpret_cast<const float2*>(p0 + DF_ROW(k) * sld);
    constexpr int U = PER >= 16 ? 1 : 16 / PER;
    for (int s0 = 1; s0 < sl_n; s0 += U) {
      float2 t[U][PER];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float* sp = p0 + (long)min(s0 + u, sl_n - 1) * sl_stride;
#pragma unroll
        for (int k = 0; k < PER; ++k) t[u][k] = *reinterpret_cast<const float2*>(sp + DF_ROW(k) * sld);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (s0 + u < sl_n) {
#pragma unroll
          for (int k = 0; k < PER; ++k) {
            v[k].x += t[u][k].x;
            v[k].y += t[u][k].y;
          }
        }
      }
    }
    float bx = 0.f, by = 0.f;
    if (sl.bias) { bx = sl.bias[c]; by = sl.bias[c + 1]; }
    if (sl.rowbias) { bx += sl.rowbias[(long)n * sl.ld_rowbias + c]; by += sl.rowbias[(long)n * sl.ld_rowbias + c + 1]; }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      v[k].x += bx;
      v[k].y += by;
    }
    if (sl_own) {       // the producer's epilogue in the reduce kernel's order (bias, then residual); x is written back
      if (sl.res) {
        float2 r[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) r[k] = *reinterpret_cast<const float2*>(sl.res + row0 * sl.ldr + c + DF_ROW(k) * sl.ldr);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
          v[k].x += r[k].x;
          v[k].y += r[k].y;
        }
      }
#pragma unroll
      for (int k = 0; k < PER; ++k)
        if (act && ty + k * R < HW) *reinterpret_cast<float2*>(sl.hout + row0 * ld + c + DF_ROW(k) * ld) = v[k];
    }
  } else {
    const float* p0 = x + row0 * ld + c;
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k] = *reinterpret_cast<const float2*>(p0 + DF_ROW(k) * ld);
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < PER; ++k)
    if (act && ty + k * R < HW) s += v[k].x + v[k].y;
  const float cnt = (float)HW * (float)cpg;
  const float mean = block_sum_dpp_fresh(s, red) / cnt;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < PER; ++k)
    if (act && ty + k * R < HW) {
      const float a = v[k].x - mean, b = v[k].y - mean;
      q += a * a + b * b;
    }
  const float rstd = rsqrtf(block_sum_dpp_fresh(q, red2) / cnt + eps);
  const float sc0 = rstd * gamma[c], sc1 = rstd * gamma[c + 1], sh0 = beta[c], sh1 = beta[c + 1];
  bf16_t* ob = out + row0 * ldo + c;
  bf16_t* rb = raw ? raw + row0 * ldo + c : nullptr;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    if (!act || ty + k * R >= HW) continue;
    float a = (v[k].x - mean) * sc0 + sh0;
    float b = (v[k].y - mean) * sc1 + sh1;
    if (silu) {
      a = silu_f(a);
      b = silu_f(b);
    }
    *reinterpret_cast<uint32_t*>(ob + DF_ROW(k) * ldo) = pack_bf2_hw(a, b);
    if (rb) *reinterpret_cast<uint32_t*>(rb + DF_ROW(k) * ldo) = pack_bf2_hw(v[k].x, v[k].y);
  }
#undef DF_ROW
}

__global__ void groupnorm_kernel(const float* __restrict__ x, int ld, int HW, int C, int cpg,
                                 const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                 int silu, bf16_t* __restrict__ out, int ldo, bf16_t* __restrict__ raw) {
  __shared__ float red[16];
  // block b runs on XCD b%8: put the 4 neighbouring groups that share 128-B lines of every pixel row on one XCD
  const int g = (blockIdx.x & 7) * 4 + (blockIdx.x >> 3), n = blockIdx.y;
  const float* xb = x + (long)n * HW * ld + g * cpg;
  const int half = cpg >> 1;              // cpg is even: float2 granularity
  const int items = HW * half;
  float s = 0.f;
  for (int i = threadIdx.x; i < items; i += blockDim.x) {
    const int px = i / half, j = i - px * half;
    const float2 v = *reinterpret_cast<const float2*>(xb + (long)px * ld + 2 * j);
    s += v.x + v.y;
  }
  const float cnt = (float)HW * (float)cpg;
  const float mean = block_sum(s, red) / cnt;
  float q = 0.f;
  for (int i = threadIdx.x; i < items; i += blockDim.x) {
    const int px = i / half, j = i - px * half;
    const float2 v = *reinterpret_cast<const float2*>(xb + (long)px * ld + 2 * j);
    const float a = v.x - mean, b = v.y - mean;
    q += a * a + b * b;
  }
  const float rstd = rsqrtf(block_sum(q, red) / cnt + eps);
  bf16_t* ob = out + (long)n * HW * ldo + g * cpg;
  bf16_t* rb = raw ? raw + (long)n * HW * ldo + g * cpg : nullptr;
  for (int i = threadIdx.x; i < items; i += blockDim.x) {
    const int px = i / half, j = i - px * half;
    const float2 v = *reinterpret_cast<const float2*>(xb + (long)px * ld + 2 * j);
    const int c = g * cpg + 2 * j;
    float a = (v.x - mean) * rstd * gamma[c] + beta[c];
    float b = (v.y - mean) * rstd * gamma[c + 1] + beta[c + 1];
    if (silu) {
      a = silu_f(a);
      b = silu_f(b);
    }
    *reinterpret_cast<uint32_t*>(ob + (long)px * ldo + 2 * j) = pack_bf2(a, b);
    if (rb) *reinterpret_cast<uint32_t*>(rb + (long)px * ldo + 2 * j) = pack_bf2(v.x, v.y);
  }
}

// ------------------------------------------------------------------ GroupNorm for LARGE slabs (VAE decoder: up to
// 128x512 px x 512 ch per sample).  The one-block-per-(group, sample) kernels above touch 40..64-B slivers of every
// pixel row and occupy 32*N blocks; here the tensor is walked in pixel chunks with ALL channels (full 512..2048-B
// rows, one float4 per thread), three launches:
//   gn_chunk_stats:  per (chunk, sample): per-group (mean, M2) of the chunk, read once into registers, two passes over the
//                    registers, written as partials -- numerically a chunked Welford, deterministic;
//   gn_merge_stats:  per (sample, group): Chan merge of the chunk partials -> (mean, rstd);
//   gn_chunk_apply:  normalise + affine [+ SiLU] -> operand type, same chunking.
// HBM traffic: 2 reads + 1 operand-type write of the tensor instead of 3 strided reads.
constexpr int GN_CHUNK_PX = 256;

// Statistics chunk: 32768 / C pixels (256 | 128 | 64 for C = 128 | 256 | 512), i.e. always 32 float4 per thread -- the chunk is
// read from memory ONCE into registers and both passes (mean, then centred squares) run on the registers (round 3; the first
// form re-read a 128..512-KB chunk, which for C = 512 no longer came out of L2: 53 us per norm, 2.5 TB/s).
__host__ __device__ inline int gn_stats_chunk_px(int C) { return 32768 / C; }
constexpr int GN_STATS_IT = 32;

__global__ __launch_bounds__(256) void gn_chunk_stats_kernel(const float* __restrict__ x, int ld, int HW, int C, int cpg,
                                                             float* __restrict__ part /*[N][chunks][32][2]*/) {
  __shared__ float red[256 * 2];
  const int n = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
  const int q = C >> 2;                       // float4 lanes per pixel (32 | 64 | 128)
  const int lane_c = threadIdx.x % q, lane_p = threadIdx.x / q, ppi = 256 / q;   // pixels per iteration
  const int cpx = gn_stats_chunk_px(C);       // == GN_STATS_IT * ppi
  const int p0 = chunk * cpx, p1 = min(HW, p0 + cpx);
  const float* xb = x + (long)n * HW * ld + lane_c * 4;
  const int qpg = max(cpg >> 2, 1);           // float4 lanes per group (cpg = 4 | 8 | 16)
  const int grp = lane_c / qpg;
  float4 v[GN_STATS_IT];
#pragma unroll
  for (int it = 0; it < GN_STATS_IT; ++it) {
    const int p = min(p0 + lane_p + it * ppi, HW - 1);           // clamped: every load is issued, tails are masked below
    v[it] = *reinterpret_cast<const float4*>(xb + (long)p * ld);
  }
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < GN_STATS_IT; ++it)
    if (p0 + lane_p + it * ppi < p1) s += (v[it].x + v[it].y) + (v[it].z + v[it].w);
  red[threadIdx.x] = s;
  __syncthreads();
  // group sum of the chunk: every thread adds the lanes of its group over all pixel lanes (<= 64 LDS reads)
  float gs = 0.f;
  for (int lp = 0; lp < ppi; ++lp)
    for (int k = 0; k < qpg; ++k) gs += red[lp * q + grp * qpg + k];
  const float cnt = (float)(p1 - p0) * (float)cpg;
  const float mean = gs / cnt;
  float m2 = 0.f;
#pragma unroll
  for (int it = 0; it < GN_STATS_IT; ++it)
    if (p0 + lane_p + it * ppi < p1) {
      const float a = v[it].x - mean, b = v[it].y - mean, c = v[it].z - mean, d = v[it].w - mean;
      m2 += (a * a + b * b) + (c * c + d * d);
    }
  __syncthreads();
  red[threadIdx.x] = m2;
  __syncthreads();
  if (lane_p == 0 && (lane_c % qpg) == 0) {
    float g2 = 0.f;
    for (int lp = 0; lp < ppi; ++lp)
      for (int k = 0; k < qpg; ++k) g2 += red[lp * q + grp * qpg + k];
    float* o = part + (((long)n * nchunk + chunk) * 32 + grp) * 2;
    o[0] = mean;
    o[1] = g2;
  }
}

__global__ __launch_bounds__(64) void gn_merge_stats_kernel(const float* __restrict__ part, int nchunk, int cpx, int HW, int cpg,
                                                            float eps, float* __restrict__ stat /*[N][32][2]*/) {
  const int n = blockIdx.y, g = blockIdx.x, lane = threadIdx.x;
  // Chan merge of the chunk partials: lane l folds chunks l, l+64, ... (independent loads), then a fixed shuffle tree
  // joins the 64 lanes -- the association order depends on nothing but nchunk, so the result is deterministic
  float cnt = 0.f, mean = 0.f, m2 = 0.f;
  for (int c = lane; c < nchunk; c += 64) {
    const float* p = part + (((long)n * nchunk + c) * 32 + g) * 2;
    const float nb = (float)(min(HW, (c + 1) * cpx) - c * cpx) * (float)cpg;
    const float d = p[0] - mean, tot = cnt + nb;
    mean += d * nb / tot;
    m2 += p[1] + d * d * cnt * nb / tot;
    cnt = tot;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float c2 = __shfl_down(cnt, o), mu2 = __shfl_down(mean, o), q2 = __shfl_down(m2, o);
    const float tot = cnt + c2;
    if (tot > 0.f) {
      const float d = mu2 - mean;
      mean += d * c2 / tot;
      m2 += q2 + d * d * cnt * c2 / tot;
    }
    cnt = tot;
  }
  if (lane == 0) {
    stat[((long)n * 32 + g) * 2] = mean;
    stat[((long)n * 32 + g) * 2 + 1] = rsqrtf(m2 / cnt + eps);
  }
}

// Round 5: a thread owns EIGHT consecutive channels (two float4 loads -> one 16-byte operand store) and keeps GN_APPLY_UN pixels in
// flight (16 independent loads per thread before the first use).  The first form walked its pixels one dependent load -> store at a
// time with 8-byte stores: 46.8 us per norm on average in the VAE decoder (4.3 TB/s on the 128 x 512 maps, where the statistics
// pass over the same bytes runs at 7.6).  Same arithmetic per element, bit-identical output.
constexpr int GN_APPLY_UN = 8;
__global__ __launch_bounds__(256) void gn_chunk_apply_kernel(const float* __restrict__ x, int ld, int HW, int C, int cpg,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ stat, int silu,
                                                             bf16_t* __restrict__ out, int ldo, bf16_t* __restrict__ raw, int chunk_px) {
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int q = C >> 3;                                 // 8-channel lanes per pixel (16 | 32 | 64)
  const int lane_c = threadIdx.x % q, lane_p = threadIdx.x / q, ppi = 256 / q;
  const int p0 = chunk * chunk_px, p1 = min(HW, p0 + chunk_px);
  const int c0 = lane_c * 8;
  float4 sc[2], sh[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = c0 + 4 * h, grp = c / cpg;
    const float mean = stat[((long)n * 32 + grp) * 2], rstd = stat[((long)n * 32 + grp) * 2 + 1];
    const float4 gm = *reinterpret_cast<const float4*>(gamma + c), bt = *reinterpret_cast<const float4*>(beta + c);
    sc[h] = make_float4(rstd * gm.x, rstd * gm.y, rstd * gm.z, rstd * gm.w);
    sh[h] = make_float4(bt.x - mean * sc[h].x, bt.y - mean * sc[h].y, bt.z - mean * sc[h].z, bt.w - mean * sc[h].w);
  }
  const float* xb = x + (long)n * HW * ld + c0;
  bf16_t* ob = out + (long)n * HW * ldo + c0;
  bf16_t* rb = raw ? raw + (long)n * HW * ldo + c0 : nullptr;
  for (int pb = p0 + lane_p; pb < p1; pb += GN_APPLY_UN * ppi) {
    float4 v[GN_APPLY_UN][2];
#pragma unroll
    for (int u = 0; u < GN_APPLY_UN; ++u) {
      const float* src = xb + (long)min(pb + u * ppi, HW - 1) * ld;     // clamped: every load is issued, tails are masked below
      v[u][0] = *reinterpret_cast<const float4*>(src);
      v[u][1] = *reinterpret_cast<const float4*>(src + 4);
    }
#pragma unroll
    for (int u = 0; u < GN_APPLY_UN; ++u) {
      const int p = pb + u * ppi;
      if (p >= p1) continue;
      uint32_t w[4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float a = v[u][h].x * sc[h].x + sh[h].x, b = v[u][h].y * sc[h].y + sh[h].y;
        float c = v[u][h].z * sc[h].z + sh[h].z, d = v[u][h].w * sc[h].w + sh[h].w;
        if (silu) { a = silu_f(a); b = silu_f(b); c = silu_f(c); d = silu_f(d); }
        w[2 * h] = pack_bf2(a, b);
        w[2 * h + 1] = pack_bf2(c, d);
      }
      *reinterpret_cast<uint4*>(ob + (long)p * ldo) = make_uint4(w[0], w[1], w[2], w[3]);
      if (rb)
        *reinterpret_cast<uint4*>(rb + (long)p * ldo) = make_uint4(pack_bf2(v[u][0].x, v[u][0].y), pack_bf2(v[u][0].z, v[u][0].w),
                                                                  pack_bf2(v[u][1].x, v[u][1].y), pack_bf2(v[u][1].z, v[u][1].w));
    }
  }
}

// ------------------------------------------------------------------ LayerNorm -> bf16, one wave per row
// C == 64 * NV exactly, so every lane issues its NV loads unconditionally (all in flight at once).
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ld, int rows,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        bf16_t* __restrict__ out) {
  constexpr int C = NV * 64;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (long)row * ld;
  float v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = xr[i * 64 + lane];
#pragma unroll
  for (int i = 0; i < NV; ++i) s += v[i];
  const float mean = wave_sum(s) * (1.0f / (float)C);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float d = v[i] - mean;
    q += d * d;
  }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / (float)C) + eps);
  bf16_t* orow = out + (long)row * C;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 64 + lane;
    orow[c] = f2bf((v[i] - mean) * rstd * gamma[c] + beta[c]);
  }
}

// ------------------------------------------------------------------ row softmax fp32 -> bf16 (one wave per row)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, bf16_t* __restrict__ p,
                                                           int rows, int T, int ldp) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* sr = s + (long)row * T;
  float m = -INFINITY;
  for (int c = lane; c < T; c += 64) m = fmaxf(m, sr[c]);
  m = wave_max(m);
  float l = 0.f;
  for (int c = lane; c < T; c += 64) l += __expf(sr[c] - m);
  l = 1.0f / wave_sum(l);
  bf16_t* pr = p + (long)row * ldp;
  for (int c = lane; c < T; c += 64) pr[c] = f2bf(__expf(sr[c] - m) * l);
  for (int c = T + lane; c < ldp; c += 64) pr[c] = (bf16_t)0;
}

// ------------------------------------------------------------------ small-M linear (time-embed MLP, emb projections,
// classifier head): one wave per output column, bf16 weights streamed once with 16-B loads, fp32 activations.
template <int MR>
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* __restrict__ a, int lda,
                                                          const bf16_t* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          int ldo, int M, int N, int K, int silu_out) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  float acc[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) acc[m] = 0.f;
  const bf16_t* wr = W + (long)n * K;
  for (int k = lane * 8; k < K; k += 64 * 8) {
    const uint4 wv = *reinterpret_cast<const uint4*>(wr + k);
    float w[8];
    w[0] = bf2f(wv.x & 0xFFFF); w[1] = bf2f(wv.x >> 16);
    w[2] = bf2f(wv.y & 0xFFFF); w[3] = bf2f(wv.y >> 16);
    w[4] = bf2f(wv.z & 0xFFFF); w[5] = bf2f(wv.z >> 16);
    w[6] = bf2f(wv.w & 0xFFFF); w[7] = bf2f(wv.w >> 16);
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      if (m < M) {
        const float4 a0 = *reinterpret_cast<const float4*>(a + (long)m * lda + k);
        const float4 a1 = *reinterpret_cast<const float4*>(a + (long)m * lda + k + 4);
        acc[m] += a0.x * w[0] + a0.y * w[1] + a0.z * w[2] + a0.w * w[3] + a1.x * w[4] + a1.y * w[5] + a1.z * w[6] +
                  a1.w * w[7];
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    const float v = wave_sum(acc[m]);
    if (lane == 0 && m < M) {
      float r = v + (bias ? bias[n] : 0.f);
      if (silu_out == 1) r = silu_f(r);
      else if (silu_out == 2) r = 1.0f / (1.0f + __expf(-r));
      out[(long)m * ldo + n] = r;
    }
  }
}

// Same contraction with the activations staged ONCE per block in LDS (M <= 16 rows x K fp32) and two weight rows per
// wave pass: the weight stream (16-B loads, read once) is the only global traffic in the loop, so the kernel runs at
// the HBM rate instead of being bound by 16 L1 activation loads per weight load.  With `tvals` the activations are the
// sinusoidal timestep embedding of tvals[m % t_B] (util.py:151-171, [cos | sin]) generated in place: the copy,
// embedding and first MLP layer of the time-embedding path are one launch.
template <int MR>
__global__ __launch_bounds__(256, MR <= 8 ? 4 : 2) void linear_rows_lds_kernel(const float* __restrict__ a, int lda,
                                                              const float* __restrict__ tvals, int t_B,
                                                              const bf16_t* __restrict__ W,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int ldo, int M, int N, int K, int act_out, int cpw) {
  extern __shared__ __attribute__((aligned(16))) float sA[];   // [MR][K]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tvals) {
    const int half = K >> 1;
    for (int i = tid; i < M * half; i += 256) {
      const int m = i / half, k = i - m * half;
      const float f = expf(-9.210340371976184f * (float)k / (float)half);
      const float ang = tvals[m % t_B] * f;
      sA[m * K + k] = cosf(ang);
      sA[m * K + half + k] = sinf(ang);
    }
  } else {
    // all of a thread's float4 loads are issued before the first LDS store (a load-store-load chain costs one L2 round
    // trip per iteration: 10 of them for 8 x 1280 activations)
    constexpr int SV = (MR * 1536 / 4 + 255) / 256;
    float4 sv[SV];
#pragma unroll
    for (int j = 0; j < SV; ++j) {
      const int i = min((tid + j * 256) * 4, M * K - 4);
      const int m = i / K, k = i - m * K;
      sv[j] = *reinterpret_cast<const float4*>(a + (long)m * lda + k);
    }
#pragma unroll
    for (int j = 0; j < SV; ++j) {
      const int i = (tid + j * 256) * 4;
      if (i < M * K) *reinterpret_cast<float4*>(&sA[i]) = sv[j];
    }
  }
  for (int i = M * K + tid; i < MR * K; i += 256) sA[i] = 0.f;
  __syncthreads();
  // Each wave owns cpw consecutive columns and walks them two at a time: 6 independent 16-B weight loads per pass
  // (K <= 1536).  The kernel is capped at 128 VGPRs (4 blocks = 16 waves per CU, LDS 4 x 40 KB), which keeps ~96 KB of
  // weight requests in flight per CU -- what it takes to stream at the HBM rate -- without any software pipelining.
  constexpr int KI = 3;
  const int col0 = (blockIdx.x * 4 + wid) * cpw;
#pragma unroll 1
  for (int pass = 0; pass < cpw / 2; ++pass) {
    const int n0 = col0 + pass * 2;
    if (n0 >= N) break;
    const bf16_t* w0 = W + (long)n0 * K;
    const bf16_t* w1 = W + (long)min(n0 + 1, N - 1) * K;
    uint4 u0[KI], u1[KI];
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int kc = min(lane * 8 + i * 512, K - 8);
      u0[i] = *reinterpret_cast<const uint4*>(w0 + kc);
      u1[i] = *reinterpret_cast<const uint4*>(w1 + kc);
    }
    float acc0[MR], acc1[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) acc0[m] = acc1[m] = 0.f;
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int k = lane * 8 + i * 512;
      if (i * 512 >= K) break;                            // wave-uniform
      const bool live = k < K;                            // lanes past K contribute nothing
      const int kc = min(k, K - 8);
      float x0[8], x1[8];
      x0[0] = bf2f(u0[i].x & 0xFFFF); x0[1] = bf2f(u0[i].x >> 16); x0[2] = bf2f(u0[i].y & 0xFFFF); x0[3] = bf2f(u0[i].y >> 16);
      x0[4] = bf2f(u0[i].z & 0xFFFF); x0[5] = bf2f(u0[i].z >> 16); x0[6] = bf2f(u0[i].w & 0xFFFF); x0[7] = bf2f(u0[i].w >> 16);
      x1[0] = bf2f(u1[i].x & 0xFFFF); x1[1] = bf2f(u1[i].x >> 16); x1[2] = bf2f(u1[i].y & 0xFFFF); x1[3] = bf2f(u1[i].y >> 16);
      x1[4] = bf2f(u1[i].z & 0xFFFF); x1[5] = bf2f(u1[i].z >> 16); x1[6] = bf2f(u1[i].w & 0xFFFF); x1[7] = bf2f(u1[i].w >> 16);
      if (!live) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x0[j] = x1[j] = 0.f;
      }
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        const float4 a0 = *reinterpret_cast<const float4*>(&sA[m * K + kc]);
        const float4 a1 = *reinterpret_cast<const float4*>(&sA[m * K + kc + 4]);
        acc0[m] += a0.x * x0[0] + a0.y * x0[1] + a0.z * x0[2] + a0.w * x0[3] + a1.x * x0[4] + a1.y * x0[5] + a1.z * x0[6] + a1.w * x0[7];
        acc1[m] += a0.x * x1[0] + a0.y * x1[1] + a0.z * x1[2] + a0.w * x1[3] + a1.x * x1[4] + a1.y * x1[5] + a1.z * x1[6] + a1.w * x1[7];
      }
    }
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const float v0 = wave_sum_dpp(acc0[m]), v1 = wave_sum_dpp(acc1[m]);
      if (lane == 0 && m < M) {
        float r0 = v0 + (bias ? bias[n0] : 0.f), r1 = v1 + (bias ? bias[min(n0 + 1, N - 1)] : 0.f);
        if (act_out == 1) { r0 = silu_f(r0); r1 = silu_f(r1); }
        else if (act_out == 2) { r0 = 1.0f / (1.0f + __expf(-r0)); r1 = 1.0f / (1.0f + __expf(-r1)); }
        out[(long)m * ldo + n0] = r0;
        if (n0 + 1 < N) out[(long)m * ldo + n0 + 1] = r1;
      }
    }
  }
}

// operand-type variant: row n uses t[n % t_B] (CFG batch duplication folded in); feeds the MFMA time-embedding MLP
__global__ void timestep_embedding_b16_kernel(const float* __restrict__ t, int t_B, bf16_t* __restrict__ out, int N, int dim) {
  const int half = dim >> 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * half) return;
  const int n = i / half, k = i - n * half;
  const float f = expf(-9.210340371976184f * (float)k / (float)half);
  const float a = t[n % t_B] * f;
  out[(long)n * dim + k] = f2bf(cosf(a));
  out[(long)n * dim + half + k] = f2bf(sinf(a));
}

__global__ void timestep_embedding_kernel(const float* __restrict__ t, float* __restrict__ out, int N, int dim) {
  const int half = dim >> 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * half) return;
  const int n = i / half, k = i - n * half;
  // freqs = exp(-ln(10000) * k / half), computed like the reference in fp32
  const float f = expf(-9.210340371976184f * (float)k / (float)half);
  const float a = t[n] * f;
  out[(long)n * dim + k] = cosf(a);
  out[(long)n * dim + half + k] = sinf(a);
}

__global__ void pack_latent_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int B, int C, int HW,
                                   int cpad, int rep, float in_scale, const float* __restrict__ wpq,
                                   const float* __restrict__ bpq) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)rep * B * HW;
  if (i >= total) return;
  const int px = (int)(i % HW);
  const int b = (int)((i / HW) % B);
  const float* xp = x + (long)b * C * HW + px;        // any C <= cpad (the reference's in_channels / z_channels are constructor arguments)
  bf16_t* o = out + i * cpad;
  if (wpq) {
    for (int oc = 0; oc < C; ++oc) {
      float s = bpq[oc];
      for (int c = 0; c < C; ++c) s += wpq[oc * C + c] * (xp[(long)c * HW] * in_scale);
      o[oc] = f2bf(s);
    }
  } else {
    for (int c = 0; c < C; ++c) o[c] = f2bf(xp[(long)c * HW] * in_scale);
  }
  for (int c = C; c < cpad; ++c) o[c] = 0;
}

// The two launches a hoisted denoise step starts with, as one (round 5): blocks [0, npack) pack the latent (pack_latent_kernel without
// the post-quant matrix), the blocks behind them broadcast row `src` of the timestep table to the `rows` rows of E (bcast_rows_kernel).
// Both are independent elementwise jobs; a launch costs ~4.5 us whatever it does.
__global__ void pack_latent_bcast_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int B, int C, int HW, int cpad,
                                         int rep, int npack, const float4* __restrict__ src, float4* __restrict__ dst, int rows,
                                         int n4) {
  if ((int)blockIdx.x < npack) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)rep * B * HW;
    if (i >= total) return;
    const int px = (int)(i % HW);
    const int b = (int)((i / HW) % B);
    bf16_t* o = out + i * cpad;
    for (int c = 0; c < C; ++c) o[c] = f2bf(x[((long)b * C + c) * HW + px] * 1.0f);
    for (int c = C; c < cpad; ++c) o[c] = 0;
    return;
  }
  const int nb = (int)gridDim.x - npack;
  for (int i = ((int)blockIdx.x - npack) * blockDim.x + threadIdx.x; i < n4; i += nb * blockDim.x) {
    const float4 v = src[i];
    for (int r = 0; r < rows; ++r) dst[(long)r * n4 + i] = v;
  }
}

__global__ void cast_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = f2bf(x[i]);
}

__global__ void cast_bf16_2d_kernel(const float* __restrict__ x, int ld, bf16_t* __restrict__ out, long rows, int C) {
  const long total = rows * (C >> 1);
  const int half = C >> 1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / half;
    const int j = (int)(i - r * half);
    const float2 v = *reinterpret_cast<const float2*>(x + r * ld + 2 * j);
    *reinterpret_cast<uint32_t*>(out + r * C + 2 * j) = pack_bf2(v.x, v.y);
  }
}

__global__ void pack_conv_weight_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int O, int I, int KH,
                                        int KW, int Ipad) {
  const long total = (long)O * KH * KW * Ipad;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(e % Ipad);
    long r = e / Ipad;
    const int kw = (int)(r % KW);
    r /= KW;
    const int kh = (int)(r % KH);
    const int o = (int)(r / KH);
    out[e] = (ci < I) ? f2bf(w[(((long)o * I + ci) * KH + kh) * KW + kw]) : (bf16_t)0;
  }
}

// conv2 of a ResBlock with its 1x1 skip connection folded in: out[o] = [ conv taps (ky,kx,ci) : 9*I | skip weights : I2 ]
__global__ void pack_conv_skip_kernel(const float* __restrict__ w, const float* __restrict__ ws, bf16_t* __restrict__ out,
                                      int O, int I, int I2) {
  const int KT = 9 * I + I2;
  const long total = (long)O * KT;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e % KT), o = (int)(e / KT);
    float v;
    if (k < 9 * I) {
      const int tap = k / I, ci = k - tap * I;
      v = w[((long)o * I + ci) * 9 + tap];
    } else {
      v = ws[(long)o * I2 + (k - 9 * I)];
    }
    out[e] = f2bf(v);
  }
}

__global__ void pack_geglu_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                  bf16_t* __restrict__ wout, float* __restrict__ bout, int half_rows, int K) {
  const long total = (long)2 * half_rows * K;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e % K);
    const int prow = (int)(e / K);               // packed row
    const int blk = prow >> 6, r = prow & 63;
    const int src = (r < 32) ? (blk * 32 + r) : (half_rows + blk * 32 + (r - 32));
    wout[e] = f2bf(w[(long)src * K + k]);
    if (k == 0) bout[prow] = b[src];
  }
}

// LayerNorm folded into the consuming Linear (attention_openai.py:211-215 pre-norms): one wave per weight row.
//   wout[dst][k] = operand(gamma[k] * w[r][k]);  cs[dst] = sum_k wout[dst][k] (as rounded);  bb[dst] = beta . w[r] + bias[r]
// dst = row_off + r, or the GEGLU (32 x | 32 gate) interleave of r when geglu_half > 0.
__global__ __launch_bounds__(256) void pack_ln_linear_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, bf16_t* __restrict__ wout,
                                                             float* __restrict__ cs, float* __restrict__ bb, int rows,
                                                             int K, int row_off, int geglu_half) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  int dst = row_off + r;
  if (geglu_half > 0) {
    const int rr = r < geglu_half ? r : r - geglu_half;
    dst = (rr >> 5) * 64 + (rr & 31) + (r < geglu_half ? 0 : 32);
  }
  float s = 0.f, b = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float wv = w[(long)r * K + k];
    const bf16_t o = f2bf(gamma[k] * wv);
    wout[(long)dst * K + k] = o;
    s += bf2f(o);
    b += beta[k] * wv;
  }
  s = wave_sum(s);
  b = wave_sum(b);
  if (lane == 0) {
    cs[dst] = s;
    bb[dst] = b + (bias ? bias[r] : 0.f);
  }
}

__global__ void cfg_combine_kernel(const float* __restrict__ e2, float* __restrict__ e, long n, float scale) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float u = e2[i], c = e2[n + i];
    e[i] = u + scale * (c - u);
  }
}

// dst[r][:] = src[:] for r < rows: one row of the hoisted time-embedding table (df_unet_set_timesteps) to every sample
__global__ void bcast_rows_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int rows, int n4) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
    const float4 v = src[i];
    for (int r = 0; r < rows; ++r) dst[(long)r * n4 + i] = v;
  }
}

struct LinArgs {
  const float* in[4];
  float coef[4];
  int n;
};
__global__ void lincomb_kernel(float* __restrict__ out, LinArgs a, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int j = 0; j < a.n; ++j) v += a.coef[j] * a.in[j][i];
    out[i] = v;
  }
}

__global__ void ddim_update_kernel(const float* __restrict__ x, const float* __restrict__ e,
                                   const float* __restrict__ noise, float* __restrict__ x_prev,
                                   float* __restrict__ pred_x0, long n, float sqrt_at, float s1m, float sqrt_aprev,
                                   float dir_coef, float sigma) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float ev = e[i];
    const float p0 = (x[i] - s1m * ev) / sqrt_at;
    float xp = sqrt_aprev * p0 + dir_coef * ev;
    if (noise) xp += sigma * noise[i];
    pred_x0[i] = p0;
    x_prev[i] = xp;
  }
}

// Inpainting blend of the samplers (ddim.py:206-209, plms.py:147-150, ddpm.py:1239-1241):
//   img <- q_sample(x0, t) * mask + (1 - mask) * img,   q_sample(x0, t) = sqrt(acp_t) x0 + sqrt(1 - acp_t) noise   (ddpm.py:279-282)
// mask is indexed modulo mask_n (1 channel broadcast over C: mask_n = H*W per sample handled by the caller's layout [B][1][H][W]).
__global__ void q_sample_blend_kernel(const float* __restrict__ img, const float* __restrict__ x0,
                                      const float* __restrict__ noise, const float* __restrict__ mask, float* __restrict__ out,
                                      long n, long chw, long hw, int mask_c, float a, float b) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long smp = i / chw, r = i - smp * chw;
    const long mi = mask_c == 1 ? smp * hw + (r % hw) : i;            // [B][1][H][W] or [B][C][H][W]
    const float m = mask[mi];
    const float q = a * x0[i] + b * noise[i];
    out[i] = q * m + (1.0f - m) * img[i];
  }
}

__global__ void avgpool_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int HW, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i - n * C;
  float s = 0.f;
  for (int p = 0; p < HW; ++p) s += x[((long)n * HW + p) * C + c];
  out[i] = s / (float)HW;
}

inline int grid_for(long n, int block = 256, int cap = 4096) {
  long g = (n + block - 1) / block;
  return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

bool groupnorm_accepts_slabs(int HW, int C) {      // shapes the register kernel takes (launch_groupnorm_sl)
  if (C % 64 != 0 || (long)HW * (C / 64) > 16384) return false;
  const int half = C / 64, rmax = std::min(HW, 1024 / half);
  return rmax > 0 && (HW + rmax - 1) / rmax <= 20;
}

size_t groupnorm_scratch_bytes(int N, int HW, int C) {
  const long items = (long)HW * (C / 64);
  if (items <= 16384 || (C != 128 && C != 256 && C != 512)) return 0;      // register kernels / generic streaming kernel
  const int cpx = gn_stats_chunk_px(C), nchunk = (HW + cpx - 1) / cpx;
  return ((size_t)N * nchunk * 32 * 2 + (size_t)N * 32 * 2) * sizeof(float);
}

hipError_t launch_groupnorm_chunked(const float* x, int ld, int N, int HW, int C, const float* gamma, const float* beta,
                                    float eps, int silu, uint16_t* out, int ldo, uint16_t* raw_out, float* scratch,
                                    hipStream_t s) {
  if ((C != 128 && C != 256 && C != 512) || (ld & 3) || (ldo & 7) || !scratch) return hipErrorInvalidValue;
  const int cpg = C / 32, cpx = gn_stats_chunk_px(C), nchunk = (HW + cpx - 1) / cpx;
  // apply chunk: 256 pixels, halved while the grid would not fill the chip twice (4 x 64 x 64 px at C = 512: 64 blocks of 256 px
  // ran the pass at 0.9 TB/s: 39 us per norm) and one chunk still feeds the eight pixels a thread keeps in flight
  int chunk_apply = GN_CHUNK_PX;
  const int ppi = 256 / (C >> 3);
  while ((long)N * ((HW + chunk_apply - 1) / chunk_apply) < 512 && chunk_apply / 2 >= GN_APPLY_UN * ppi) chunk_apply /= 2;
  const int nchunk_apply = (HW + chunk_apply - 1) / chunk_apply;
  float* part = scratch;
  float* stat = scratch + (size_t)N * nchunk * 32 * 2;
  hipLaunchKernelGGL(gn_chunk_stats_kernel, dim3(nchunk, N), dim3(256), 0, s, x, ld, HW, C, cpg, part);
  hipLaunchKernelGGL(gn_merge_stats_kernel, dim3(32, N), dim3(64), 0, s, part, nchunk, cpx, HW, cpg, eps, stat);
  hipLaunchKernelGGL(gn_chunk_apply_kernel, dim3(nchunk_apply, N), dim3(256), 0, s, x, ld, HW, C, cpg, gamma, beta, stat, silu,
                     out, ldo, raw_out, chunk_apply);
  return hipGetLastError();
}

static hipError_t launch_groupnorm_sl(const float* x, int ld, int N, int HW, int C, const float* gamma, const float* beta,
                                      float eps, int silu, uint16_t* out, int ldo, uint16_t* raw_out, const GnSlabs& sl,
                                      hipStream_t s);

hipError_t launch_groupnorm_slabs(const float* x, int ld, int N, int HW, int C, const float* gamma, const float* beta,
                                  float eps, int silu, uint16_t* out, int ldo, uint16_t* raw_out, int nslab,
                                  long slab_stride, const float* bias, const float* rowbias, int ld_rowbias, hipStream_t s) {
  GnSlabs sl{nslab, slab_stride, bias, rowbias, ld_rowbias, nullptr, 0, nullptr, 0, nullptr};
  return launch_groupnorm_sl(x, ld, N, HW, C, gamma, beta, eps, silu, out, ldo, raw_out, sl, s);
}

hipError_t launch_groupnorm_own_slabs(float* x, int ld, int N, int HW, int C, const float* gamma, const float* beta, float eps,
                                      int silu, uint16_t* out, int ldo, uint16_t* raw_out, const float* slab0, int nslab,
                                      long slab_stride, int c_own, const float* bias, const float* res, int ldr, hipStream_t s) {
  if (nslab < 2 || !slab0 || c_own <= 0 || c_own > C || (c_own & 1) || (ld & 1) || (ldr & 1)) return hipErrorInvalidValue;
  GnSlabs sl{nslab, slab_stride, bias, nullptr, 0, slab0, c_own, res, ldr, x};
  return launch_groupnorm_sl(x, ld, N, HW, C, gamma, beta, eps, silu, out, ldo, raw_out, sl, s);
}

static hipError_t launch_groupnorm_sl(const float* x, int ld, int N, int HW, int C, const float* gamma, const float* beta,
                                      float eps, int silu, uint16_t* out, int ldo, uint16_t* raw_out, const GnSlabs& sl,
                                      hipStream_t s) {
  if (C % 64 != 0) return hipErrorInvalidValue;
  const int cpg = C / 32;
  const long items = (long)HW * (cpg / 2);
  const int nslab = sl.n;
  if (nslab > 0 && items > 16384) return hipErrorInvalidValue;     // the streaming kernel has no slab path
  // thread layout of groupnorm_reg_kernel: R rows per pass x (cpg / 2) channel pairs, PER passes
  const int half = cpg / 2;
  const int rmax = std::min(HW, 1024 / std::max(half, 1));
  const int need = rmax > 0 ? (HW + rmax - 1) / rmax : 1 << 30;
#define DF_GN_REG(PER)                                                                                                \
  {                                                                                                                   \
    const int R = (HW + (PER) - 1) / (PER);                                                                           \
    const int threads = (half * R + 63) & ~63;                                                                        \
    hipLaunchKernelGGL(groupnorm_reg_kernel<PER>, dim3(32 * N), dim3(threads), 0, s, x, ld, HW, cpg, R, sl.n, N, sl.own, \
                       sl.c_own, sl.stride, out, ldo, silu, C, gamma, beta, eps, raw_out, sl);                        \
  }
  if (items > 16384 || need > 20 || half < 1) {
    if (nslab > 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(groupnorm_kernel, dim3(32, N), dim3(1024), 0, s, x, ld, HW, C, cpg, gamma, beta, eps, silu, out,
                       ldo, raw_out);
  } else if (need <= 1) DF_GN_REG(1)
  else if (need <= 2) DF_GN_REG(2)
  else if (need <= 3) DF_GN_REG(3)
  else if (need <= 4) DF_GN_REG(4)
  else if (need <= 6) DF_GN_REG(6)
  else if (need <= 8) DF_GN_REG(8)
  else if (need <= 12) DF_GN_REG(12)
  else if (need <= 16) DF_GN_REG(16)
  else DF_GN_REG(20)
#undef DF_GN_REG
  return hipGetLastError();
}

hipError_t launch_groupnorm(const float* x, int ld, int N, int HW, int C, const float* gamma, const float* beta,
                            float eps, int silu, uint16_t* out, int ldo, uint16_t* raw_out, hipStream_t s) {
  return launch_groupnorm_slabs(x, ld, N, HW, C, gamma, beta, eps, silu, out, ldo, raw_out, 0, 0, nullptr, nullptr, 0, s);
}

hipError_t launch_layernorm(const float* x, int ld, int rows, int C, const float* gamma, const float* beta,
                            float eps, uint16_t* out, hipStream_t s) {
  const int blocks = (rows + 3) / 4;
#define DF_LN(NV) hipLaunchKernelGGL(layernorm_kernel<NV>, dim3(blocks), dim3(256), 0, s, x, ld, rows, gamma, beta, eps, out)
  // one instantiation per row length C = 64 NV, NV = 1 .. 32 (the Stage-2 UNet meets 320 / 640 / 1280 and the classifier 128 / 256 /
  // 512; other model_channels x channel_mult products -- 192, 384, 768, 1024 ... -- reach this kernel on 1- / 2- / 3-token maps)
  if (C <= 0 || C % 64 != 0 || C > 2048) return hipErrorInvalidValue;
#define DF_LN4(N0) case N0: DF_LN(N0); break; case N0 + 1: DF_LN(N0 + 1); break; case N0 + 2: DF_LN(N0 + 2); break; \
                   case N0 + 3: DF_LN(N0 + 3); break;
  switch (C / 64) {
    DF_LN4(1) DF_LN4(5) DF_LN4(9) DF_LN4(13) DF_LN4(17) DF_LN4(21) DF_LN4(25) DF_LN4(29)
    default: return hipErrorInvalidValue;
  }
#undef DF_LN4
#undef DF_LN
  return hipGetLastError();
}

hipError_t launch_softmax_rows(const float* sc, uint16_t* p, int rows, int T, int ldp, hipStream_t st) {
  if (ldp < T) return hipErrorInvalidValue;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, sc, p, rows, T, ldp);
  return hipGetLastError();
}

hipError_t launch_linear_rows(const float* a, int lda, const uint16_t* W, const float* bias, float* out, int ldo,
                              int M, int N, int K, int silu_out, hipStream_t s) {
  if (K % 8 != 0 || lda % 4 != 0) return hipErrorInvalidValue;
  const int blocks = (N + 3) / 4;
  for (int m0 = 0; m0 < M; m0 += 16) {   // chunks of 16 rows (weights re-streamed per chunk; M is tiny here)
    const int mm = (M - m0) < 16 ? (M - m0) : 16;
    if (mm <= 4)
      hipLaunchKernelGGL(linear_rows_kernel<4>, dim3(blocks), dim3(256), 0, s, a + (long)m0 * lda, lda, W, bias,
                         out + (long)m0 * ldo, ldo, mm, N, K, silu_out);
    else if (mm <= 8)
      hipLaunchKernelGGL(linear_rows_kernel<8>, dim3(blocks), dim3(256), 0, s, a + (long)m0 * lda, lda, W, bias,
                         out + (long)m0 * ldo, ldo, mm, N, K, silu_out);
    else
      hipLaunchKernelGGL(linear_rows_kernel<16>, dim3(blocks), dim3(256), 0, s, a + (long)m0 * lda, lda, W, bias,
                         out + (long)m0 * ldo, ldo, mm, N, K, silu_out);
  }
  return hipGetLastError();
}

hipError_t launch_linear_rows_lds(const float* a, int lda, const float* tvals, int t_B, const uint16_t* W,
                                  const float* bias, float* out, int ldo, int M, int N, int K, int act_out, hipStream_t s) {
  if (K % 8 != 0 || K < 8 || K > 1536 || (a && lda % 4 != 0) || M < 1 || M > 16) return hipErrorInvalidValue;
  const int cpw = N >= 16384 ? 6 : (N >= 4096 ? 4 : 2);      // columns per wave: all blocks co-resident on 256 CUs
  const int blocks = (N + 4 * cpw - 1) / (4 * cpw);
#define DF_LRL(MR)                                                                                                  \
  {                                                                                                                 \
    const size_t lds = (size_t)MR * K * 4;                                                                          \
    static size_t attr = 0;                                                                                         \
    if (lds > attr) {                                                                                               \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_rows_lds_kernel<MR>),                \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
      if (e != hipSuccess) return e;                                                                                \
      attr = lds;                                                                                                   \
    }                                                                                                               \
    hipLaunchKernelGGL(linear_rows_lds_kernel<MR>, dim3(blocks), dim3(256), lds, s, a, lda, tvals, t_B, W, bias, out, ldo, \
                       M, N, K, act_out, cpw);                                                                      \
  }
  if (M <= 4) DF_LRL(4)
  else if (M <= 8) DF_LRL(8)
  else DF_LRL(16)
#undef DF_LRL
  return hipGetLastError();
}

hipError_t launch_timestep_embedding_b16(const float* t, int t_B, uint16_t* out, int N, int dim, hipStream_t s) {
  const int n = N * (dim / 2);
  hipLaunchKernelGGL(timestep_embedding_b16_kernel, dim3((n + 255) / 256), dim3(256), 0, s, t, t_B, out, N, dim);
  return hipGetLastError();
}

hipError_t launch_timestep_embedding(const float* t, float* out, int N, int dim, hipStream_t s) {
  const int n = N * (dim / 2);
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0, s, t, out, N, dim);
  return hipGetLastError();
}

hipError_t launch_pack_latent(const float* x, uint16_t* out, int B, int C, int HW, int cpad, int rep, float in_scale,
                              const float* wpq, const float* bpq, hipStream_t s) {
  if (C > cpad || C <= 0) return hipErrorInvalidValue;
  const long n = (long)rep * B * HW;
  hipLaunchKernelGGL(pack_latent_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, s, x, out, B, C, HW, cpad, rep,
                     in_scale, wpq, bpq);
  return hipGetLastError();
}

hipError_t launch_pack_latent_bcast(const float* x, uint16_t* out, int B, int C, int HW, int cpad, int rep, const float* src,
                                    float* dst, int rows, int n, hipStream_t s) {
  if (C > cpad || C <= 0 || n <= 0 || (n & 3) != 0 || rows <= 0) return hipErrorInvalidValue;
  const long np = (long)rep * B * HW;
  const int npack = (int)((np + 255) / 256), nb = grid_for(n / 4);
  hipLaunchKernelGGL(pack_latent_bcast_kernel, dim3(npack + nb), dim3(256), 0, s, x, out, B, C, HW, cpad, rep, npack,
                     (const float4*)src, (float4*)dst, rows, n / 4);
  return hipGetLastError();
}

hipError_t launch_cast_bf16(const float* x, uint16_t* out, long n, hipStream_t s) {
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, out, n);
  return hipGetLastError();
}

hipError_t launch_cast_bf16_2d(const float* x, int ld, uint16_t* out, long rows, int C, hipStream_t s) {
  if (C % 2 != 0 || ld % 2 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(cast_bf16_2d_kernel, dim3(grid_for(rows * (C / 2))), dim3(256), 0, s, x, ld, out, rows, C);
  return hipGetLastError();
}

// nearest-x2 upsample + conv3x3 == four 2x2-tap convs on the input-resolution map (output phase (a, b) = (Y & 1, X & 1)): the
// taps of the 3x3 kernel that land on the same input pixel are summed in fp32 and rounded ONCE:
//   W'[a][b][o][dy][dx][c] = sum_{ky in S_a(dy)} sum_{kx in S_b(dx)} w[o][c][ky][kx],  S_0 = ({0}, {1, 2}),  S_1 = ({0, 1}, {2});
// tap (dy, dx) reads input pixel (y - 1 + a + dy, x - 1 + b + dx).  out: [4][O][4][Ipad] operand type.
__global__ void pack_conv_ups4_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int O, int I, int Ipad) {
  const long total = (long)16 * O * Ipad;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(e % Ipad);
    long r = e / Ipad;
    const int tap = (int)(r & 3);
    r >>= 2;
    const int o = (int)(r % O), ph = (int)(r / O);
    const int a = ph >> 1, b = ph & 1, dy = tap >> 1, dx = tap & 1;
    const int ky0 = a == 0 ? (dy == 0 ? 0 : 1) : (dy == 0 ? 0 : 2), ky1 = a == 0 ? (dy == 0 ? 0 : 2) : (dy == 0 ? 1 : 2);
    const int kx0 = b == 0 ? (dx == 0 ? 0 : 1) : (dx == 0 ? 0 : 2), kx1 = b == 0 ? (dx == 0 ? 0 : 2) : (dx == 0 ? 1 : 2);
    float acc = 0.f;
    if (ci < I)
      for (int ky = ky0; ky <= ky1; ++ky)
        for (int kx = kx0; kx <= kx1; ++kx) acc += w[(((long)o * I + ci) * 3 + ky) * 3 + kx];
    out[e] = f2bf(acc);
  }
}

hipError_t launch_pack_conv_ups4(const float* w, uint16_t* out, int O, int I, int Ipad, hipStream_t s) {
  const long n = (long)16 * O * Ipad;
  hipLaunchKernelGGL(pack_conv_ups4_kernel, dim3(grid_for(n)), dim3(256), 0, s, w, out, O, I, Ipad);
  return hipGetLastError();
}

namespace {
// 3x3 convolution (stride 1, zero padding 1) with a HANDFUL of output channels straight to an NCHW fp32 tensor: the VAE decoder's
// conv_out (stage1_autoencoder/model.py: Decoder.conv_out, 128 -> 3 channels over the full 128 x 512 mel image).  On the implicit-GEMM
// tiles the 3 output channels are padded to a 64-column tile (21x the MFMA work, 144 us at B = 4); the op is bound by reading the
// activations once.  A block owns a 4-row x 32-pixel patch: its 6 x 34 halo pixels are staged into LDS with fully coalesced 16-byte
// loads (16 lanes = one pixel's channel vector; a first form that loaded the MFMA fragments straight from global memory -- lane =
// pixel, 32 B used of every 128-byte line per instruction -- ran at 133 us, bound by the texture-address path), zeros outside the
// image.  Wavefront w owns row w of the patch: A operand = its 32 pixels' channel vectors out of LDS (row pitch C * 2 + 16 B: the 16
// lanes of a read phase hit 64 distinct banks), B operand = the weight rows of the <= 4 output channels out of LDS (lanes >= Cout hold
// zeros), one v_mfma_f32_32x32x16 per 16 input channels and tap.  Lane (co = lane & 31 < Cout, half) ends up with 16 pixels of output
// channel co: four float4 stores.
constexpr int FO_TH = 4, FO_TW = 32;
__global__ __launch_bounds__(256) void conv3x3_fewout_kernel(const bf16_t* __restrict__ X /*[NB][H][W][C]*/, const bf16_t* __restrict__ Wt /*[Cout][9][C]*/,
                                                             const float* __restrict__ bias, float* __restrict__ out /*[NB][Cout][H][W]*/,
                                                             int NB, int H, int W, int C, int Cout) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int pitch = C * 2 + 16;                                  // bytes per staged pixel
  char* sX = smem;                                               // [(FO_TH + 2) * (FO_TW + 2)][pitch]
  bf16_t* sW = reinterpret_cast<bf16_t*>(smem + (FO_TH + 2) * (FO_TW + 2) * pitch);      // [Cout][9 * C]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int tpr = (W + FO_TW - 1) / FO_TW, tpc = (H + FO_TH - 1) / FO_TH;
  const int bx = blockIdx.x % tpr, by = (blockIdx.x / tpr) % tpc, n = blockIdx.x / (tpr * tpc);
  const int x0 = bx * FO_TW, y0 = by * FO_TH;
  // ---- stage the halo patch: chunk = 16 bytes, C / 8 chunks per pixel
  const int cpp = C >> 3, nchunks = (FO_TH + 2) * (FO_TW + 2) * cpp;
  for (int i = tid; i < nchunks; i += 256) {
    const int px = i / cpp, ch = i - px * cpp;
    const int py = px / (FO_TW + 2), pxx = px - py * (FO_TW + 2);
    const int yy = y0 + py - 1, xx = x0 + pxx - 1;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = *reinterpret_cast<const uint4*>(X + (((long)n * H + yy) * W + xx) * C + ch * 8);
    *reinterpret_cast<uint4*>(sX + px * pitch + ch * 16) = v;
  }
  for (int i = tid; i < Cout * 9 * C / 8; i += 256) reinterpret_cast<uint4*>(sW)[i] = reinterpret_cast<const uint4*>(Wt)[i];
  __syncthreads();
  const int y = y0 + wid;
  if (y >= H) return;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int nch = C >> 4;
  const bool wrow = l31 < Cout;
  const bf16_t* wl = sW + (wrow ? l31 : 0) * 9 * C + lh * 8;
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {
    const int dy = tap / 3, dx = tap - dy * 3;                   // patch coordinates: (wid + dy, l31 + dx)
    const char* src = sX + ((wid + dy) * (FO_TW + 2) + l31 + dx) * pitch + lh * 16;
    const bf16_t* wt = wl + tap * C;
#pragma unroll 8
    for (int c = 0; c < nch; ++c) {
      const uint4 av = *reinterpret_cast<const uint4*>(src + c * 32);
      uint4 bv = *reinterpret_cast<const uint4*>(wt + c * 16);
      if (!wrow) bv = make_uint4(0, 0, 0, 0);
      uint4 a2 = av;
      acc = DF_MFMA_32x32x16(*reinterpret_cast<bf16x8*>(&a2), *reinterpret_cast<bf16x8*>(&bv), acc);
    }
  }
  if (wrow) {
    const float b = bias ? bias[l31] : 0.f;
    float* o = out + (((long)n * Cout + l31) * H + y) * W + x0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int px = 8 * g + 4 * lh;
      if (x0 + px + 3 < W) {
        *reinterpret_cast<float4*>(o + px) = make_float4(acc[4 * g] + b, acc[4 * g + 1] + b, acc[4 * g + 2] + b, acc[4 * g + 3] + b);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (x0 + px + i < W) o[px + i] = acc[4 * g + i] + b;
      }
    }
  }
#endif
}
}  // namespace

static size_t conv3x3_fewout_lds(int C, int Cout) { return (size_t)(FO_TH + 2) * (FO_TW + 2) * (C * 2 + 16) + (size_t)Cout * 9 * C * 2; }

bool conv3x3_fewout_ok(int H, int W, int C, int Cout) {
  return Cout >= 1 && Cout <= 4 && C % 16 == 0 && C >= 16 && conv3x3_fewout_lds(C, Cout) <= 160 * 1024 && W % 4 == 0 && H > 0;
}

hipError_t launch_conv3x3_fewout(const uint16_t* X, const uint16_t* Wt, const float* bias, float* out, int NB, int H, int W, int C,
                                 int Cout, hipStream_t s) {
  if (!conv3x3_fewout_ok(H, W, C, Cout)) return hipErrorInvalidValue;
  const long blocks = (long)NB * ((H + FO_TH - 1) / FO_TH) * ((W + FO_TW - 1) / FO_TW);
  const size_t lds = conv3x3_fewout_lds(C, Cout);
  static size_t attr = 0;
  if (lds > attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_fewout_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr = lds;
  }
  hipLaunchKernelGGL(conv3x3_fewout_kernel, dim3((unsigned)blocks), dim3(256), lds, s, X, Wt, bias, out, NB, H, W, C, Cout);
  return hipGetLastError();
}

hipError_t launch_pack_conv_weight(const float* w, uint16_t* out, int O, int I, int KH, int KW, int Ipad,
                                   hipStream_t s) {
  const long n = (long)O * KH * KW * Ipad;
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(grid_for(n)), dim3(256), 0, s, w, out, O, I, KH, KW, Ipad);
  return hipGetLastError();
}

namespace {
// FeedForward's second Linear followed by proj_out (attention_openai.py:86-98, 261) is ONE linear map of (h, t):
//   proj_out(t + W2 h + b2) = (Wp W2) h + Wp t + (Wp b2 + bp)
// packed as the operand [C][4C | C] = [Wp.W2 | Wp] (fp32 product, rounded once) and the fp32 bias Wp.b2 + bp.
__global__ __launch_bounds__(256) void pack_ffproj_kernel(const float* __restrict__ Wp /*[C][C]*/,
                                                          const float* __restrict__ W2 /*[C][F]*/, bf16_t* __restrict__ out /*[C][F+C]*/,
                                                          int C, int F) {
  __shared__ float sa[32][33], sb[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  if (j0 >= F) {                                   // the Wp columns: plain operand-type copy
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(long)(i0 + ty + 8 * r) * (F + C) + j0 + tx] = f2bf(Wp[(long)(i0 + ty + 8 * r) * C + (j0 - F) + tx]);
    return;
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < C; k0 += 32) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sa[ty + 8 * r][tx] = Wp[(long)(i0 + ty + 8 * r) * C + k0 + tx];
      sb[ty + 8 * r][tx] = W2[(long)(k0 + ty + 8 * r) * F + j0 + tx];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) {
      const float b = sb[kk][tx];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] += sa[ty + 8 * r][kk] * b;
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) out[(long)(i0 + ty + 8 * r) * (F + C) + j0 + tx] = f2bf(acc[r]);
}
__global__ __launch_bounds__(256) void ffproj_bias_kernel(const float* __restrict__ Wp, const float* __restrict__ b2,
                                                          const float* __restrict__ bp, float* __restrict__ out, int C) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // one wavefront per output
  if (i >= C) return;
  float a = 0.f;
  for (int k = lane; k < C; k += 64) a += Wp[(long)i * C + k] * b2[k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
  if (lane == 0) out[i] = a + bp[i];
}
}  // namespace

hipError_t launch_pack_ffproj(const float* Wp, const float* bp, const float* W2, const float* b2, uint16_t* wout, float* bout,
                              int C, int F, hipStream_t s) {
  if (C % 32 != 0 || F % 32 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pack_ffproj_kernel, dim3((F + C) / 32, C / 32), dim3(256), 0, s, Wp, W2, wout, C, F);
  hipLaunchKernelGGL(ffproj_bias_kernel, dim3((C + 3) / 4), dim3(256), 0, s, Wp, b2, bp, bout, C);
  return hipGetLastError();
}

namespace {
// ---- cross-attention with the context folded into per-sample "weights" (engine.hip: context_px).
// kv [NB*Tc][2C] = (K | V) of the context tokens.  Kexp / Vexp [NB][H*Tcp][C]: row (h, tc) holds the head-h slice of token
// tc's K (V) in columns [h*D, (h+1)*D) and zeros elsewhere, so that ONE dense GEMM against a C x C weight yields every head's
// (K_h Wq_h) resp. (Wo_h V_h^T) block.  Rows tc >= Tc are padding (zero).
__global__ __launch_bounds__(256) void xattn_expand_kernel(const bf16_t* __restrict__ kv, bf16_t* __restrict__ Kexp,
                                                           bf16_t* __restrict__ Vexp, int Tc, int Tcp, int C, int H, long total8) {
  const int D = C / H, c8n = C / 8;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total8; e += (long)gridDim.x * 256) {
    const int j8 = (int)(e % c8n);
    const long row = e / c8n;                      // (n, h, tp)
    const int tp = (int)(row % Tcp), h = (int)((row / Tcp) % H);
    const long n = row / ((long)Tcp * H);
    uint4 k4 = make_uint4(0, 0, 0, 0), v4 = make_uint4(0, 0, 0, 0);
    if (tp < Tc) {
      const bf16_t* src = kv + ((n * Tc + tp) * 2l * C) + j8 * 8;
      k4 = *reinterpret_cast<const uint4*>(src);
      v4 = *reinterpret_cast<const uint4*>(src + C);
      // 8 consecutive columns may straddle a head boundary when D % 8 != 0 is impossible here: D is a multiple of 8
      const int hj = (j8 * 8) / D;
      if (hj != h) { k4 = make_uint4(0, 0, 0, 0); v4 = k4; }
    }
    *reinterpret_cast<uint4*>(Kexp + row * C + j8 * 8) = k4;
    *reinterpret_cast<uint4*>(Vexp + row * C + j8 * 8) = v4;
  }
}
// WqT[c][j] = scale * gamma[c] * Wq[j][c]  (operand type): the LayerNorm-folded query projection, transposed, as the
// "weight" operand of  G' = Kexp . WqT^T  (G'[row][c] = scale * sum_j Kexp[row][j] gamma[c] Wq[j][c])
__global__ __launch_bounds__(256) void pack_lnq_t_kernel(const float* __restrict__ Wq, const float* __restrict__ gamma,
                                                         bf16_t* __restrict__ out, int C, float scale) {
  __shared__ float t[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int j0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
#pragma unroll
  for (int r = 0; r < 4; ++r) t[ty + 8 * r][tx] = Wq[(long)(j0 + ty + 8 * r) * C + c0 + tx];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = c0 + ty + 8 * r;
    out[(long)c * C + j0 + tx] = f2bf(scale * gamma[c] * t[tx][ty + 8 * r]);
  }
}
// per row of G' [rows][C]: cs = sum_c G'[row][c] (of the operand-rounded values the MFMA multiplies) and
// bb = scale * sum_j Kexp[row][j] * bq[j]  with bq = Wq . beta (the LayerNorm shift through the query projection)
__global__ __launch_bounds__(256) void xattn_rowstats_kernel(const bf16_t* __restrict__ G, const bf16_t* __restrict__ Kexp,
                                                             const float* __restrict__ bq, float scale, int C, long rows,
                                                             float* __restrict__ cs, float* __restrict__ bb) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float a = 0.f, b = 0.f;
  for (int c = lane; c < C; c += 64) {
    a += bf2f(G[row * C + c]);
    b += bf2f(Kexp[row * C + c]) * bq[c];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
  if (lane == 0) { cs[row] = a; bb[row] = scale * b; }
}
}  // namespace

hipError_t launch_xattn_expand(const uint16_t* kv, uint16_t* Kexp, uint16_t* Vexp, int NB, int Tc, int Tcp, int C, int H,
                               hipStream_t s) {
  if (C % H != 0 || (C / H) % 8 != 0 || C % 8 != 0) return hipErrorInvalidValue;
  const long total8 = (long)NB * H * Tcp * (C / 8);
  const int blocks = (int)std::min<long>((total8 + 255) / 256, 8192);
  hipLaunchKernelGGL(xattn_expand_kernel, dim3(blocks), dim3(256), 0, s, kv, Kexp, Vexp, Tc, Tcp, C, H, total8);
  return hipGetLastError();
}
hipError_t launch_pack_lnq_t(const float* Wq, const float* gamma, uint16_t* out, int C, float scale, hipStream_t s) {
  if (C % 32 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pack_lnq_t_kernel, dim3(C / 32, C / 32), dim3(256), 0, s, Wq, gamma, out, C, scale);
  return hipGetLastError();
}
hipError_t launch_xattn_rowstats(const uint16_t* G, const uint16_t* Kexp, const float* bq, float scale, int C, long rows, float* cs,
                                 float* bb, hipStream_t s) {
  hipLaunchKernelGGL(xattn_rowstats_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, G, Kexp, bq, scale, C, rows, cs, bb);
  return hipGetLastError();
}

hipError_t launch_pack_conv_skip(const float* w, const float* ws, uint16_t* out, int O, int I, int I2, hipStream_t s) {
  const long n = (long)O * (9 * I + I2);
  hipLaunchKernelGGL(pack_conv_skip_kernel, dim3(grid_for(n)), dim3(256), 0, s, w, ws, out, O, I, I2);
  return hipGetLastError();
}

hipError_t launch_pack_geglu(const float* w, const float* b, uint16_t* wout, float* bout, int half_rows, int K,
                             hipStream_t s) {
  if (half_rows % 32 != 0) return hipErrorInvalidValue;
  const long n = (long)2 * half_rows * K;
  hipLaunchKernelGGL(pack_geglu_kernel, dim3(grid_for(n)), dim3(256), 0, s, w, b, wout, bout, half_rows, K);
  return hipGetLastError();
}

hipError_t launch_pack_ln_linear(const float* w, const float* bias, const float* gamma, const float* beta,
                                 uint16_t* wout, float* cs, float* bb, int rows, int K, int row_off, int geglu_half,
                                 hipStream_t s) {
  if (geglu_half > 0 && (geglu_half % 32 != 0 || rows != 2 * geglu_half || row_off != 0)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pack_ln_linear_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, w, bias, gamma, beta, wout, cs, bb, rows,
                     K, row_off, geglu_half);
  return hipGetLastError();
}

hipError_t launch_cfg_combine(const float* e2, float* e, long n, float scale, hipStream_t s) {
  hipLaunchKernelGGL(cfg_combine_kernel, dim3(grid_for(n)), dim3(256), 0, s, e2, e, n, scale);
  return hipGetLastError();
}

hipError_t launch_bcast_rows(const float* src, float* dst, int rows, int n, hipStream_t s) {
  if (n <= 0 || (n & 3) != 0 || rows <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bcast_rows_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, (const float4*)src, (float4*)dst, rows, n / 4);
  return hipGetLastError();
}

hipError_t launch_lincomb(float* out, const float* const* in, const float* coef, int nterms, long n, hipStream_t s) {
  if (nterms < 1 || nterms > 4) return hipErrorInvalidValue;
  LinArgs a;
  a.n = nterms;
  for (int i = 0; i < 4; ++i) {
    a.in[i] = i < nterms ? in[i] : nullptr;
    a.coef[i] = i < nterms ? coef[i] : 0.f;
  }
  hipLaunchKernelGGL(lincomb_kernel, dim3(grid_for(n)), dim3(256), 0, s, out, a, n);
  return hipGetLastError();
}

hipError_t launch_q_sample_blend(const float* img, const float* x0, const float* noise, const float* mask, float* out, long n,
                                 long chw, long hw, int mask_c, float a, float b, hipStream_t s) {
  if (n <= 0 || chw <= 0 || hw <= 0 || n % chw != 0 || chw % hw != 0 || (mask_c != 1 && mask_c != (int)(chw / hw)))
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(q_sample_blend_kernel, dim3(grid_for(n)), dim3(256), 0, s, img, x0, noise, mask, out, n, chw, hw, mask_c, a, b);
  return hipGetLastError();
}

hipError_t launch_ddim_update(const float* x, const float* e, const float* noise, float* x_prev, float* pred_x0,
                              long n, float sqrt_at, float s1m, float sqrt_aprev, float dir_coef, float sigma,
                              hipStream_t s) {
  hipLaunchKernelGGL(ddim_update_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, e, noise, x_prev, pred_x0, n, sqrt_at,
                     s1m, sqrt_aprev, dir_coef, sigma);
  return hipGetLastError();
}

hipError_t launch_avgpool(const float* x, float* out, int N, int HW, int C, hipStream_t s) {
  hipLaunchKernelGGL(avgpool_kernel, dim3((N * C + 255) / 256), dim3(256), 0, s, x, out, N, HW, C);
  return hipGetLastError();
}
