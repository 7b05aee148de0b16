// Mel -> waveform on the GPU: the `inverse_op` tail of the notebook (inference/demo_util.py:196-211) --
// undo the log-mel normalisation, invert the mel filterbank by non-negative least squares (librosa mel_to_stft), then 32
// iterations of fast Griffin-Lim (librosa griffinlim: momentum 0.99, hann window, n_fft 1024, hop 256, centred frames).
// In the reference this is 24 of the notebook's 30 seconds (CPU, librosa).  Everything here is fp32 VALU work on small
// per-frame problems: HBM/L2-bound streaming plus 1024-point FFTs held in LDS; no MFMA (no GEMM-shaped reuse to speak of:
// the 128 x 513 filterbank is the only matrix and it is applied to 16-frame tiles out of L2).
//
//   nnls_fista_kernel   per tile of 16 frames: x >= 0 minimising |A x - b|^2 by FISTA (projected gradient with Nesterov
//                       momentum, step 1/L), started from the clipped least-squares solution like librosa; x, the
//                       momentum point and the residual live in LDS; both products walk only the non-zero spans of the
//                       banded mel filterbank (~1000 of 65 664 entries).
//                       (librosa drives the same objective with L-BFGS-B; the minimiser of this under-determined problem
//                       is not unique, so the two agree in the residual, not element by element -- oracle/vocoder.py.)
//   gl_ifft_kernel      one frame per block: spectrum = S * angles, Hermitian extension, radix-2 FFT in LDS, window.
//   gl_ola_kernel       window-sum-square normalised overlap-add (gather form: deterministic), centre trimmed.
//   gl_stft_kernel      reflect-padded frame * window -> FFT -> angle update  a = rebuilt - m/(1+m) * previous; a /= |a|.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int NFFT = 1024, NBIN = NFFT / 2 + 1, HOPV = 256, TT = 16, FPAD = 516;
constexpr int NNLS_NT = 1024;      // 16 wavefronts per 16-frame tile: the iteration is a chain of short dependent loops, latency-bound

// ---- in-LDS radix-2 FFT of 1024 complex points, 256 threads; input already in bit-reversed order; tw[k] = exp(-2 pi i k / 1024)
__device__ __forceinline__ void fft1024(float2* s, const float2* __restrict__ tw, int tid, float sgn) {
#pragma unroll 1
  for (int len = 2, stride = NFFT / 2; len <= NFFT; len <<= 1, stride >>= 1) {
    const int half = len >> 1;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = tid + q * 256;
      const int j = i & (half - 1), a = ((i - j) << 1) + j, b = a + half;
      float2 w = tw[j * stride];
      w.y *= sgn;
      const float2 u = s[a], v = s[b];
      // scalar butterfly: as packed expressions the complex product and the +- pair become `v_pk_mul_f32 ... op_sel:[1,1]` /
      // `v_pk_add_f32 ... op_sel:[0,1]` -- src1 operand selects (common.h pk_add_hi, tools/check_pk_opsel.py).  Every result goes
      // through its own register barrier so that the vectoriser finds no pair to pack.
      float tx = w.x * v.x, ty = w.x * v.y;
      asm volatile("" : "+v"(tx));
      asm volatile("" : "+v"(ty));
      tx = fmaf(-w.y, v.y, tx);
      ty = fmaf(w.y, v.x, ty);
      float ax = u.x + tx, ay = u.y + ty, bx = u.x - tx, by = u.y - ty;
      asm volatile("" : "+v"(ax));
      asm volatile("" : "+v"(ay));
      asm volatile("" : "+v"(bx));
      asm volatile("" : "+v"(by));
      s[a] = make_float2(ax, ay);
      s[b] = make_float2(bx, by);
    }
  }
  __syncthreads();
}
__device__ __forceinline__ int brev10(int i) { return (int)(__brev((unsigned)i) >> 22); }

// ---- NNLS by FISTA on tiles of TT frames.  mel [B][NM][T] (normalised log-mel) ; A [NM][F] ; At [F][NM] ; Pt [NM][F] = pinv(A)^T
__global__ __launch_bounds__(NNLS_NT) void nnls_fista_kernel(const float* __restrict__ mel, int NM, int T, const float* __restrict__ A,
                                                         const float* __restrict__ At, const float* __restrict__ Pt,
                                                         float inv_L, int iters, float* __restrict__ S /*[B][T][NBIN]*/) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* x = sm;                          // [FPAD][TT]
  float* yv = x + FPAD * TT;              // [FPAD][TT]   momentum point
  float* r = yv + FPAD * TT;              // [NM][TT]     residual A y - b
  float* bm = r + NM * TT;                // [NM][TT]     linear-amplitude mel of this tile
  const int tid = threadIdx.x, t0 = blockIdx.x * TT, b = blockIdx.y;
  const float* melb = mel + (long)b * NM * T;
  for (int e = tid; e < NM * TT; e += NNLS_NT) {
    const int m = e / TT, tt = e - m * TT;
    const int t = min(t0 + tt, T - 1);
    const float v = melb[(long)m * T + t];
    bm[e] = exp10f((v * 100.f - 100.f + 20.f) / 20.f);        // spec*100-100 ; (spec+20)/20 ; 10**spec
  }
  __syncthreads();
  // x0 = max(pinv(A) b, 0)
  for (int k = tid; k < NBIN; k += NNLS_NT) {
    float acc[TT];
#pragma unroll
    for (int j = 0; j < TT; ++j) acc[j] = 0.f;
    for (int m = 0; m < NM; ++m) {
      const float p = Pt[(long)m * NBIN + k];
#pragma unroll
      for (int j = 0; j < TT; ++j) acc[j] += p * bm[m * TT + j];
    }
#pragma unroll
    for (int j = 0; j < TT; ++j) x[k * TT + j] = yv[k * TT + j] = fmaxf(acc[j], 0.f);
  }
  // The mel filterbank is banded: a triangular filter touches 2..16 neighbouring FFT bins and a bin belongs to at most two
  // filters, i.e. ~1000 of the 128 x 513 entries are non-zero.  Every block derives the non-zero span of each row and each
  // column from A itself (any matrix works; a dense one just has full spans) and both products of an iteration walk spans
  // only: 33x less work than the dense products, same summation order over the non-zeros (bit-identical result).
  __shared__ short klo[128], khi[128], mlo[FPAD], mhi[FPAD];
  {
    const int lane = tid & 63, w = tid >> 6;
    for (int m = w; m < NM; m += NNLS_NT / 64) {                     // one wavefront per row, coalesced along k
      int lo = NBIN, hi = 0;
      for (int k = lane; k < NBIN; k += 64)
        if (A[(long)m * NBIN + k] != 0.f) { lo = min(lo, k); hi = max(hi, k + 1); }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
      if (lane == 0) { klo[m] = (short)min(lo, hi); khi[m] = (short)hi; }
    }
    for (int k = tid; k < NBIN; k += NNLS_NT) {               // one thread per column, coalesced across threads
      int lo = NM, hi = 0;
      for (int m = 0; m < NM; ++m)
        if (A[(long)m * NBIN + k] != 0.f) { lo = min(lo, m); hi = m + 1; }
      mlo[k] = (short)min(lo, hi); mhi[k] = (short)hi;
    }
  }
  float tn = 1.f;
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
    for (int e = tid; e < NM * TT; e += NNLS_NT) {            // residual r = A y - b over the row spans
      const int m = e / TT, j = e - m * TT;
      float acc = 0.f;
      const float* Am = A + (long)m * NBIN;
      for (int k = klo[m]; k < khi[m]; ++k) acc += Am[k] * yv[k * TT + j];
      r[e] = acc - bm[e];
    }
    __syncthreads();
    const float tnext = 0.5f * (1.f + sqrtf(1.f + 4.f * tn * tn));
    const float beta = (tn - 1.f) / tnext;
    for (int e = tid; e < NBIN * TT; e += NNLS_NT) {          // gradient A^T r over the column spans, projected step, momentum
      const int k = e / TT, j = e - k * TT;
      float g = 0.f;
      const float* Ak = At + (long)k * NM;
      for (int m = mlo[k]; m < mhi[k]; ++m) g += Ak[m] * r[m * TT + j];
      const float xo = x[e];
      const float xn = fmaxf(yv[e] - inv_L * g, 0.f);
      x[e] = xn;
      yv[e] = xn + beta * (xn - xo);
    }
    tn = tnext;
  }
  __syncthreads();
  float* Sb = S + ((long)b * T + t0) * NBIN;
  for (int e = tid; e < TT * NBIN; e += NNLS_NT) {
    const int tt = e / NBIN, k = e - tt * NBIN;
    if (t0 + tt < T) Sb[(long)tt * NBIN + k] = x[k * TT + tt];
  }
}

// ---- Griffin-Lim pieces.  S fp32 [B][T][NBIN]; angles / rebuilt complex [B][T][NBIN]; frames fp32 [B][T][NFFT]
__global__ __launch_bounds__(256) void gl_init_angles_kernel(const float* __restrict__ phase0 /*[B][NBIN][T] in [0,1)*/,
                                                             float2* __restrict__ angles, int T, long total) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int k = (int)(e % NBIN);
    const long bt = e / NBIN;
    const int t = (int)(bt % T);
    const long b = bt / T;
    float s, c;
    sincospif(2.0f * phase0[(b * NBIN + k) * T + t], &s, &c);      // exp(2 pi i u)
    angles[e] = make_float2(c, s);
  }
}

__global__ __launch_bounds__(256) void gl_ifft_kernel(const float* __restrict__ S, const float2* __restrict__ angles,
                                                      const float2* __restrict__ tw, const float* __restrict__ window,
                                                      float* __restrict__ frames) {
  __shared__ float2 s[NFFT];
  const int tid = threadIdx.x;
  const long f = blockIdx.x;                       // frame index b*T + t
  const float* Sf = S + f * NBIN;
  const float2* af = angles + f * NBIN;
  for (int k = tid; k < NBIN; k += 256) {
    const float mag = Sf[k];
    float2 v = make_float2(mag * af[k].x, mag * af[k].y);
    if (k == 0 || k == NFFT / 2) v.y = 0.f;        // irfft ignores the imaginary part of DC / Nyquist
    s[brev10(k)] = v;
    if (k > 0 && k < NFFT / 2) s[brev10(NFFT - k)] = make_float2(v.x, -v.y);
  }
  fft1024(s, tw, tid, -1.f);                       // conj twiddles: inverse transform
  float* o = frames + f * NFFT;
  for (int n = tid; n < NFFT; n += 256) o[n] = s[n].x * (1.0f / NFFT) * window[n];
}

__global__ __launch_bounds__(256) void gl_ola_kernel(const float* __restrict__ frames, const float* __restrict__ wss, int T, int L,
                                                     float* __restrict__ y /*[B][L]*/, int B) {
  const long total = (long)B * L;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int n = (int)(e % L);
    const long b = e / L;
    const int pos = n + NFFT / 2;                  // position in the untrimmed overlap-add signal
    const int thi = min(pos / HOPV, T - 1), tlo = max((pos - NFFT + HOPV) / HOPV, 0);
    float acc = 0.f;
    for (int t = tlo; t <= thi; ++t) acc += frames[((long)b * T + t) * NFFT + (pos - t * HOPV)];
    const float w = wss[pos];
    y[e] = (w > 1.17549435e-38f) ? acc / w : acc;
  }
}

__global__ __launch_bounds__(256) void gl_stft_kernel(const float* __restrict__ y, int T, int L, const float2* __restrict__ tw,
                                                      const float* __restrict__ window, const float2* __restrict__ prev,
                                                      float2* __restrict__ rebuilt, float2* __restrict__ angles, float mom) {
  __shared__ float2 s[NFFT];
  const int tid = threadIdx.x;
  const long f = blockIdx.x;
  const int t = (int)(f % T);
  const float* yb = y + (f / T) * L;
  for (int j = tid; j < NFFT; j += 256) {
    int i = t * HOPV + j - NFFT / 2;               // np.pad(mode="reflect"): no edge repeat
    if (i < 0 || i >= L) {                         // clips shorter than the 512-sample pad (T <= 3) reflect more than once: the
      const int P = 2 * (L - 1);                   // padded signal is the triangular wave of period 2 (L - 1), as numpy builds it
      i = P > 0 ? i % P : 0;
      if (i < 0) i += P;
      if (i >= L) i = P - i;
    }
    s[brev10(j)] = make_float2(yb[i] * window[j], 0.f);
  }
  fft1024(s, tw, tid, 1.f);
  for (int k = tid; k < NBIN; k += 256) {
    const float2 rb = s[k];
    const float2 pv = prev[f * NBIN + k];
    float2 a = make_float2(rb.x - mom * pv.x, rb.y - mom * pv.y);
    const float inv = 1.0f / (sqrtf(a.x * a.x + a.y * a.y) + 1e-16f);
    rebuilt[f * NBIN + k] = rb;
    angles[f * NBIN + k] = make_float2(a.x * inv, a.y * inv);
  }
}

}  // namespace

hipError_t launch_mel_to_stft(const float* mel, int B, int NM, int T, const float* A, const float* At, const float* Pt,
                              float inv_L, int iters, float* S, hipStream_t s) {
  if (NM > 128 || NM < 1 || T < 1) return hipErrorInvalidValue;
  const size_t lds = ((size_t)2 * FPAD * TT + (size_t)2 * NM * TT) * sizeof(float);
  static size_t attr = 0;
  if (lds > attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&nnls_fista_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr = lds;
  }
  hipLaunchKernelGGL(nnls_fista_kernel, dim3((T + TT - 1) / TT, B), dim3(NNLS_NT), lds, s, mel, NM, T, A, At, Pt, inv_L, iters, S);
  return hipGetLastError();
}

hipError_t launch_griffinlim(const float* S, const float* phase0, int B, int T, int n_iter, float momentum, const float2* tw,
                             const float* window, const float* wss, float2* angles, float2* reb0, float2* reb1, float* frames,
                             float* y, hipStream_t s) {
  if (T < 2) return hipErrorInvalidValue;
  const int L = HOPV * (T - 1);
  const long nspec = (long)B * T * NBIN;
  const int gb = (int)std::min<long>((nspec + 255) / 256, 65535), gy = (int)std::min<long>(((long)B * L + 255) / 256, 65535);
  hipLaunchKernelGGL(gl_init_angles_kernel, dim3(gb), dim3(256), 0, s, phase0, angles, T, nspec);
  hipError_t e = hipMemsetAsync(reb0, 0, (size_t)nspec * sizeof(float2), s);          // rebuilt = 0.0 before the first iteration
  if (e != hipSuccess) return e;
  const float mom = momentum / (1.0f + momentum);
  float2 *prev = reb0, *cur = reb1;
  for (int it = 0; it < n_iter; ++it) {
    hipLaunchKernelGGL(gl_ifft_kernel, dim3(B * T), dim3(256), 0, s, S, angles, tw, window, frames);
    hipLaunchKernelGGL(gl_ola_kernel, dim3(gy), dim3(256), 0, s, frames, wss, T, L, y, B);
    hipLaunchKernelGGL(gl_stft_kernel, dim3(B * T), dim3(256), 0, s, y, T, L, tw, window, prev, cur, angles, mom);
    std::swap(prev, cur);
  }
  hipLaunchKernelGGL(gl_ifft_kernel, dim3(B * T), dim3(256), 0, s, S, angles, tw, window, frames);
  hipLaunchKernelGGL(gl_ola_kernel, dim3(gy), dim3(256), 0, s, frames, wss, T, L, y, B);
  return hipGetLastError();
}
