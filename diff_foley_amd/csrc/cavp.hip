// CAVP video encoder (SlowOnly-R50): the data-movement kernels around the MFMA GEMMs.
// Activations are frame-major NHWC: [F = B*T frames][H][W][C], operand type (bf16/fp16) for GEMM inputs, fp32 for the
// residual stream.  Reference semantics: inference/model/cavp_modules.py:167-330 (Bottleneck3d), :757-779 (stem),
// :837-859 (forward); inference/model/cavp_model.py:47-65 (encode_video).
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace {

inline int grid_for(long n, int block = 256, int cap = 65535) {
  long g = (n + block - 1) / block;
  return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

// Stem im2col: fp32 NCHW frames [F][3][H][W] -> operand rows [F*OH*OW][KP] for the (1,7,7) stride-2 pad-3 conv:
// k = (ky*7 + kx)*3 + c for k < 147, zeros up to KP (= 192, a multiple of the GEMM's K step).
__global__ void stem_im2col_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int F, int H, int W, int OH,
                                   int OW, int KP) {
  const long total = (long)F * OH * OW * (KP / 2);
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int k2 = (int)(e % (KP / 2));
    const long row = e / (KP / 2);
    const int ox = (int)(row % OW), oy = (int)((row / OW) % OH), f = (int)(row / ((long)OW * OH));
    float v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = 2 * k2 + h;
      float val = 0.f;
      if (k < 147) {
        const int tap = k / 3, c = k - tap * 3, ky = tap / 7, kx = tap - ky * 7;
        const int iy = oy * 2 - 3 + ky, ix = ox * 2 - 3 + kx;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) val = x[(((long)f * 3 + c) * H + iy) * W + ix];
      }
      v[h] = val;
    }
    *reinterpret_cast<uint32_t*>(out + row * KP + 2 * k2) = pack_bf2(v[0], v[1]);
  }
}

// MaxPool (1,3,3) stride (1,2,2) pad (0,1,1) on operand-type NHWC [F][H][W][C] -> [F][OH][OW][C]; 8 channels / thread.
__global__ void maxpool3x3s2_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int F, int H, int W, int OH,
                                    int OW, int C) {
  const int C8 = C / 8;
  const long total = (long)F * OH * OW * C8;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(e % C8);
    const long px = e / C8;
    const int ox = (int)(px % OW), oy = (int)((px / OW) % OH), f = (int)(px / ((long)OW * OH));
    float m[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = -3.0e38f;
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(x + (((long)f * H + iy) * W + ix) * C + c8 * 8);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          m[2 * i] = fmaxf(m[2 * i], bf2f((uint16_t)(w[i] & 0xFFFF)));
          m[2 * i + 1] = fmaxf(m[2 * i + 1], bf2f((uint16_t)(w[i] >> 16)));
        }
      }
    }
    uint4 o;
    o.x = pack_bf2(m[0], m[1]); o.y = pack_bf2(m[2], m[3]); o.z = pack_bf2(m[4], m[5]); o.w = pack_bf2(m[6], m[7]);
    *reinterpret_cast<uint4*>(out + px * C + c8 * 8) = o;
  }
}

// Spatial subsample by 2 (the rows a 1x1 stride-2 conv reads): [F][H][W][C] -> [F][H/2][W/2][C], 16-byte chunks.
__global__ void subsample2_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int F, int H, int W, int C) {
  const int C8 = C / 8, OH = H / 2, OW = W / 2;
  const long total = (long)F * OH * OW * C8;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(e % C8);
    const long px = e / C8;
    const int ox = (int)(px % OW), oy = (int)((px / OW) % OH), f = (int)(px / ((long)OW * OH));
    *reinterpret_cast<uint4*>(out + px * C + c8 * 8) =
        *reinterpret_cast<const uint4*>(x + (((long)f * H + 2 * oy) * W + 2 * ox) * C + c8 * 8);
  }
}

// Temporal K-concatenation for the inflated (3,1,1) pad (1,0,0) convs: out[f][p][dt*C + c] = x[f + dt - 1][p][c] with
// zeros outside the clip (frames are grouped in clips of T).  One GEMM with K = 3C then evaluates the temporal conv.
__global__ void tcat3_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int F, int T, int HW, int C) {
  const int C8 = C / 8;
  const long total = (long)F * HW * 3 * C8;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(e % C8);
    const int dt = (int)((e / C8) % 3);
    const long row = e / (3 * C8);
    const int f = (int)(row / HW), t = f % T, ts = t + dt - 1;
    uint4 v = make_uint4(0, 0, 0, 0);
    if ((unsigned)ts < (unsigned)T) v = *reinterpret_cast<const uint4*>(x + (row + (long)(dt - 1) * HW) * C + c8 * 8);
    *reinterpret_cast<uint4*>(out + row * 3 * C + dt * C + c8 * 8) = v;
  }
}

// Conv3d weight [O][I][KT][KH][KW] with eval-mode BatchNorm folded in:
//   out[o][((kt*KH + ky)*KW + kx)*I + i] = w * gamma[o] / sqrt(var[o] + eps)   (operand type, zero padded to KP columns)
//   bias[o] = beta[o] - mean[o] * gamma[o] / sqrt(var[o] + eps)
__global__ void pack_conv3d_bn_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, const float* __restrict__ mean,
                                      const float* __restrict__ var, float eps, bf16_t* __restrict__ out,
                                      float* __restrict__ bias, int O, int I, int KT, int KH, int KW, int KP) {
  const long total = (long)O * KP;
  const int K = KT * KH * KW * I;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e % KP), o = (int)(e / KP);
    const float sc = gamma[o] * rsqrtf(var[o] + eps);
    float v = 0.f;
    if (k < K) {
      const int i = k % I, tap = k / I;              // tap = (kt*KH + ky)*KW + kx  == offset inside [KT][KH][KW]
      v = w[((long)o * I + i) * (KT * KH * KW) + tap] * sc;
    }
    out[e] = f2bf(v);
    if (k == 0) bias[o] = beta[o] - mean[o] * sc;
  }
}

// rows /= max(||row||_2, 1e-12)   (F.normalize(dim=-1)); one wavefront per row.
__global__ void l2norm_rows_kernel(float* __restrict__ x, int rows, int C) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = x[(long)row * C + c];
    s += v * v;
  }
  s = wave_sum(s);
  const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
  for (int c = lane; c < C; c += 64) x[(long)row * C + c] *= inv;
}


// ------------------------------------------------------------------ frame pre-processing (demo_util.py:100-104, 150-151)
// torchvision Resize((h, w)) on a PIL image = PIL.Image.resize(BILINEAR): separable, antialiased (triangle filter whose
// support grows with the down-scaling factor), 8-bit fixed point exactly as Pillow's Resample.c: horizontal pass into a
// uint8 intermediate, then the vertical pass (the other way round on shrinking frames with H > 100 W, below); every value is (2^21 + sum pixel * coeff) >> 22, clipped to 0..255.  The
// 22-bit coefficient tables are computed on the host in double precision (diff_foley_amd/video.py).  The vertical pass
// also does ToTensor(): HWC uint8 -> CHW float / 255.  Results are bit-identical to Pillow.
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ tmp, long rows,
                                                       int W, int OW, const int* __restrict__ bounds,
                                                       const int* __restrict__ coef, int ksize) {
  const long total = rows * OW * 3;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % 3);
    const int ox = (int)((e / 3) % OW);
    const long row = e / (3L * OW);
    const int x0 = bounds[2 * ox], n = bounds[2 * ox + 1];
    const uint8_t* src = in + (row * W + x0) * 3 + c;
    const int* k = coef + (long)ox * ksize;
    int acc = 1 << 21;
    for (int x = 0; x < n; ++x) acc += (int)src[3 * x] * k[x];
    acc >>= 22;
    tmp[e] = (uint8_t)min(max(acc, 0), 255);
  }
}

__global__ __launch_bounds__(256) void resize_v_totensor_kernel(const uint8_t* __restrict__ tmp, float* __restrict__ out,
                                                                int T, int H, int OH, int OW,
                                                                const int* __restrict__ bounds,
                                                                const int* __restrict__ coef, int ksize) {
  const long total = (long)T * 3 * OH * OW;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(e % OW);
    const int oy = (int)((e / OW) % OH);
    const int c = (int)((e / ((long)OW * OH)) % 3);
    const int t = (int)(e / (3L * OW * OH));
    const int y0 = bounds[2 * oy], n = bounds[2 * oy + 1];
    const uint8_t* src = tmp + (((long)t * H + y0) * OW + ox) * 3 + c;
    const int* k = coef + (long)oy * ksize;
    int acc = 1 << 21;
    for (int y = 0; y < n; ++y) acc += (int)src[(long)y * OW * 3] * k[y];
    acc >>= 22;
    out[e] = (float)min(max(acc, 0), 255) / 255.0f;
  }
}

// Pass order.  Pillow runs the horizontal pass first -- except on frames more than 100 times taller than wide (H > 100 W) whose height
// shrinks (OH < H), where it runs the vertical pass first (observed on Pillow 12.2.0: the switch sits at exactly H = 100 W + 1 and at
// OH = H - 1 whatever the output width; ~1000 random geometries against PIL, tests/test_video_cpu.py walks both boundaries).  The uint8
// rounding of the intermediate makes the two orders differ by one count in up to 10 % of the values, so the order is part of the result.
// No real video has that shape; the kernels below exist for bit parity.
__global__ __launch_bounds__(256) void resize_v_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ tmp, int T, int H, int W,
                                                       int OH, const int* __restrict__ bounds, const int* __restrict__ coef,
                                                       int ksize) {
  const long total = (long)T * OH * W * 3;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long xc = e % (3L * W);                      // (x, c) inside a row
    const int oy = (int)((e / (3L * W)) % OH);
    const int t = (int)(e / (3L * W * OH));
    const int y0 = bounds[2 * oy], n = bounds[2 * oy + 1];
    const uint8_t* src = in + ((long)t * H + y0) * W * 3 + xc;
    const int* k = coef + (long)oy * ksize;
    int acc = 1 << 21;
    for (int y = 0; y < n; ++y) acc += (int)src[(long)y * W * 3] * k[y];
    acc >>= 22;
    tmp[e] = (uint8_t)min(max(acc, 0), 255);
  }
}

__global__ __launch_bounds__(256) void resize_h_totensor_kernel(const uint8_t* __restrict__ tmp, float* __restrict__ out, int T,
                                                                int W, int OH, int OW, const int* __restrict__ bounds,
                                                                const int* __restrict__ coef, int ksize) {
  const long total = (long)T * 3 * OH * OW;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(e % OW);
    const int oy = (int)((e / OW) % OH);
    const int c = (int)((e / ((long)OW * OH)) % 3);
    const int t = (int)(e / (3L * OW * OH));
    const int x0 = bounds[2 * ox], n = bounds[2 * ox + 1];
    const uint8_t* src = tmp + (((long)t * OH + oy) * W + x0) * 3 + c;
    const int* k = coef + (long)ox * ksize;
    int acc = 1 << 21;
    for (int x = 0; x < n; ++x) acc += (int)src[3 * x] * k[x];
    acc >>= 22;
    out[e] = (float)min(max(acc, 0), 255) / 255.0f;
  }
}

}  // namespace

hipError_t launch_stem_im2col(const float* x, uint16_t* out, int F, int H, int W, int OH, int OW, int KP, hipStream_t s) {
  if (KP < 148 || (KP & 1)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(stem_im2col_kernel, dim3(grid_for((long)F * OH * OW * (KP / 2))), dim3(256), 0, s, x, out, F, H, W,
                     OH, OW, KP);
  return hipGetLastError();
}
hipError_t launch_maxpool3x3s2(const uint16_t* x, uint16_t* out, int F, int H, int W, int OH, int OW, int C, hipStream_t s) {
  if (C % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for((long)F * OH * OW * (C / 8))), dim3(256), 0, s, x, out, F, H, W, OH,
                     OW, C);
  return hipGetLastError();
}
hipError_t launch_subsample2(const uint16_t* x, uint16_t* out, int F, int H, int W, int C, hipStream_t s) {
  if (C % 8 || (H & 1) || (W & 1)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(subsample2_kernel, dim3(grid_for((long)F * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0, s, x, out, F, H,
                     W, C);
  return hipGetLastError();
}
hipError_t launch_tcat3(const uint16_t* x, uint16_t* out, int F, int T, int HW, int C, hipStream_t s) {
  if (C % 8 || T <= 0 || F % T) return hipErrorInvalidValue;
  hipLaunchKernelGGL(tcat3_kernel, dim3(grid_for((long)F * HW * 3 * (C / 8))), dim3(256), 0, s, x, out, F, T, HW, C);
  return hipGetLastError();
}
hipError_t launch_pack_conv3d_bn(const float* w, const float* gamma, const float* beta, const float* mean,
                                 const float* var, float eps, uint16_t* out, float* bias, int O, int I, int KT, int KH,
                                 int KW, int KP, hipStream_t s) {
  if (KP < KT * KH * KW * I) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pack_conv3d_bn_kernel, dim3(grid_for((long)O * KP)), dim3(256), 0, s, w, gamma, beta, mean, var, eps,
                     out, bias, O, I, KT, KH, KW, KP);
  return hipGetLastError();
}
// nn.MaxPool1d(kernel_size = k) over the frame axis of [B][T][C] features (stride k, floor: T / k windows), cavp_model.py:31, 58-59
__global__ void maxpool_time_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int T, int C, int k) {
  const int To = T / k;
  const long n = (long)B * To * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long bt = i / C;
    const int to = (int)(bt % To), b = (int)(bt / To);
    const float* p = x + ((long)b * T + (long)to * k) * C + c;
    float m = p[0];
    for (int j = 1; j < k; ++j) m = fmaxf(m, p[(long)j * C]);
    out[i] = m;
  }
}

hipError_t launch_maxpool_time(const float* x, float* out, int B, int T, int C, int k, hipStream_t s) {
  if (B <= 0 || C <= 0 || k <= 0 || T < k) return hipErrorInvalidValue;
  const long n = (long)B * (T / k) * C;
  hipLaunchKernelGGL(maxpool_time_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 4096)), dim3(256), 0, s, x, out, B, T, C, k);
  return hipGetLastError();
}

hipError_t launch_l2norm_rows(float* x, int rows, int C, hipStream_t s) {
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, rows, C);
  return hipGetLastError();
}

hipError_t launch_frames_to_tensor(const uint8_t* frames, float* out, uint8_t* tmp, int T, int H, int W, int OH, int OW,
                                   const int* bounds_w, const int* coef_w, int ksize_w, const int* bounds_h,
                                   const int* coef_h, int ksize_h, hipStream_t s) {
  if (T <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || ksize_w <= 0 || ksize_h <= 0) return hipErrorInvalidValue;
  const bool v_first = (long)H > 100L * W && OH < H;          // Pillow's pass order (see resize_v_kernel)
  const long n1 = v_first ? (long)T * OH * W * 3 : (long)T * H * OW * 3, n2 = (long)T * 3 * OH * OW;
  const int g1 = (int)std::min<long>((n1 + 255) / 256, 65535), g2 = (int)std::min<long>((n2 + 255) / 256, 65535);
  if (v_first) {
    hipLaunchKernelGGL(resize_v_kernel, dim3(g1), dim3(256), 0, s, frames, tmp, T, H, W, OH, bounds_h, coef_h, ksize_h);
    hipLaunchKernelGGL(resize_h_totensor_kernel, dim3(g2), dim3(256), 0, s, tmp, out, T, W, OH, OW, bounds_w, coef_w, ksize_w);
  } else {
    hipLaunchKernelGGL(resize_h_kernel, dim3(g1), dim3(256), 0, s, frames, tmp, (long)T * H, W, OW, bounds_w, coef_w, ksize_w);
    hipLaunchKernelGGL(resize_v_totensor_kernel, dim3(g2), dim3(256), 0, s, tmp, out, T, H, OH, OW, bounds_h, coef_h, ksize_h);
  }
  return hipGetLastError();
}
