// Host-side launchers of the non-GEMM kernels (norms, attention, small linears, sampler updates).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// GroupNorm(32 groups) [+ SiLU] over an fp32 NHWC tensor [N][HW][ld] -> bf16 [N][HW][ldo].
// raw_out (optional) receives the un-normalised input cast to bf16 (operand of a 1x1 skip conv).
hipError_t launch_groupnorm(const float* x, int ld, int N, int HW, int C, const float* gamma, const float* beta,
                            float eps, int silu, uint16_t* out, int ldo, uint16_t* raw_out, hipStream_t s);

// Same with a split-K producer: x = first fp32 partial slab [rows][ld] of the conv that feeds this norm; the kernel sums
// nslab slabs (slab_stride floats apart) and adds the conv's bias / per-sample bias on the fly (nslab = 0: plain input).
hipError_t launch_groupnorm_slabs(const float* x, int ld, int N, int HW, int C, const float* gamma, const float* beta,
                                  float eps, int silu, uint16_t* out, int ldo, uint16_t* raw_out, int nslab,
                                  long slab_stride, const float* bias, const float* rowbias, int ld_rowbias, hipStream_t s);
bool groupnorm_accepts_slabs(int HW, int C);
// x is the NOT-YET-REDUCED output of its own split-K producer: channels [0, c_own) are summed from nslab slabs (leading
// dimension c_own), bias and the fp32 residual `res` are added in the reduce kernel's order and the result is written back
// to x before it is normalised; channels [c_own, C) are read from x (the skip half of a concat buffer).
hipError_t launch_groupnorm_own_slabs(float* x, int ld, int N, int HW, int C, const float* gamma, const float* beta, float eps,
                                      int silu, uint16_t* out, int ldo, uint16_t* raw_out, const float* slab0, int nslab,
                                      long slab_stride, int c_own, const float* bias, const float* res, int ldr, hipStream_t s);

// Large slabs (VAE decoder, C in {128, 256, 512}, more than 16384 float2 per group): pixel-chunked three-launch form with
// fully coalesced rows; `scratch` holds groupnorm_scratch_bytes(N, HW, C) bytes (0 = shape not handled: use launch_groupnorm).
size_t groupnorm_scratch_bytes(int N, int HW, int C);
hipError_t launch_groupnorm_chunked(const float* x, int ld, int N, int HW, int C, const float* gamma, const float* beta,
                                    float eps, int silu, uint16_t* out, int ldo, uint16_t* raw_out, float* scratch,
                                    hipStream_t s);

// LayerNorm over the last dim of fp32 [rows][ld] -> bf16 [rows][C]   (eps 1e-5, affine)
hipError_t launch_layernorm(const float* x, int ld, int rows, int C, const float* gamma, const float* beta,
                            float eps, uint16_t* out, hipStream_t s);

// Fused attention  O = softmax(scale * Q K^T) V  per (batch, head).
//   Q  bf16 [N*Tq][ldq]  head h at columns [h*D, (h+1)*D)
//   K  bf16 [N*Tk][ldk]  same head layout
//   Vt bf16 [N][heads*D][ldvt]  (V transposed: keys contiguous), ldvt >= round_up(Tk, 32) and zero padded
//   O  bf16 [N*Tq][ldo]
hipError_t launch_attention(const uint16_t* Q, int ldq, const uint16_t* K, int ldk, const uint16_t* Vt, int ldvt,
                            uint16_t* O, int ldo, int N, int heads, int D, int Tq, int Tk, float scale,
                            hipStream_t s);
bool attention_supported(int D);

// 3x3 convolution with <= 4 output channels, NHWC operand-type input -> NCHW fp32 output (elementwise.hip; the VAE decoder's conv_out)
bool conv3x3_fewout_ok(int H, int W, int C, int Cout);
hipError_t launch_conv3x3_fewout(const uint16_t* X, const uint16_t* Wt /*[Cout][9][C]*/, const float* bias, float* out, int NB, int H, int W,
                                 int C, int Cout, hipStream_t s);

// Row softmax: fp32 scores [rows][T] -> bf16 probabilities [rows][T]  (VAE single-head attention)
// p rows have ldp >= T elements; columns [T, ldp) are written as zeros (the K padding of the P V GEMM)
hipError_t launch_softmax_rows(const float* s, uint16_t* p, int rows, int T, int ldp, hipStream_t st);

// out[m][n] = act_out( sum_k a[m][k] * W[n][k] + bias[n] ),  M <= 16, fp32 activations, bf16 weights.
hipError_t launch_linear_rows(const float* a, int lda, const uint16_t* W, const float* bias, float* out, int ldo,
                              int M, int N, int K, int act_out /*0 none, 1 silu, 2 sigmoid*/, hipStream_t s);

// Same, activations staged in LDS (weight stream is the only global traffic; HBM-rate).  tvals != nullptr: the
// activations are the sinusoidal embedding (dim K) of tvals[m % t_B], generated in the kernel (`a` is ignored).
hipError_t launch_linear_rows_lds(const float* a, int lda, const float* tvals, int t_B, const uint16_t* W,
                                  const float* bias, float* out, int ldo, int M, int N, int K, int act_out, hipStream_t s);

// Sinusoidal timestep embedding [cos | sin] (util.py:151-171), t fp32 (may be fractional) -> [N][dim] fp32
hipError_t launch_timestep_embedding(const float* t, float* out, int N, int dim, hipStream_t s);

// Same as operand-type rows [N][dim]; row n embeds t[n % t_B] (CFG batch duplication)
hipError_t launch_timestep_embedding_b16(const float* t, int t_B, uint16_t* out, int N, int dim, hipStream_t s);

// NCHW fp32 latent -> NHWC bf16 padded to cpad channels.  `rep` copies of the batch are written back to
// back (rep=2 builds the CFG batch cat([x, x])).  Optional affine pre-1x1 conv (VAE post_quant_conv):
// y = Wpq * (x * in_scale) + bpq.
hipError_t launch_pack_latent(const float* x, uint16_t* out, int B, int C, int HW, int cpad, int rep, float in_scale,
                              const float* wpq, const float* bpq, hipStream_t s);

// fp32 -> bf16 cast of a contiguous buffer (n elements, n % 2 == 0 not required)
hipError_t launch_cast_bf16(const float* x, uint16_t* out, long n, hipStream_t s);

// fp32 [rows][ld] (first C columns) -> bf16 [rows][C]
hipError_t launch_cast_bf16_2d(const float* x, int ld, uint16_t* out, long rows, int C, hipStream_t s);

// Weight re-pack on device: conv OIHW fp32 -> [O][kh][kw][Ipad] bf16 ; (Ipad >= I, zero filled)
hipError_t launch_pack_conv_weight(const float* w, uint16_t* out, int O, int I, int KH, int KW, int Ipad, hipStream_t s);
// conv3x3 OIHW fp32 -> per-phase 2x2-tap weights [4][O][4][Ipad] of the nearest-x2-upsample + conv3x3 pair (gemm_m3.hip)
hipError_t launch_pack_conv_ups4(const float* w, uint16_t* out, int O, int I, int Ipad, hipStream_t s);
// GEGLU weight/bias interleave: rows [x(4C) ; gate(4C)] -> blocks of (32 x-rows | 32 gate-rows)
// LayerNorm-folded Linear weight (one call per stacked matrix): operand rows gamma*W at row_off (or GEGLU-interleaved
// when geglu_half > 0), their column sums cs and the folded bias bb = beta.W + bias
hipError_t launch_pack_ln_linear(const float* w, const float* bias, const float* gamma, const float* beta,
                                 uint16_t* wout, float* cs, float* bb, int rows, int K, int row_off, int geglu_half,
                                 hipStream_t s);
// conv3x3 OIHW + 1x1 skip weight [O][I2] -> operand rows [O][9*I + I2] (taps (ky,kx,ci), then the skip channels)
hipError_t launch_pack_conv_skip(const float* w, const float* ws, uint16_t* out, int O, int I, int I2, hipStream_t s);
// ---- cross-attention with the context folded into per-sample weights (see engine.hip context_px)
hipError_t launch_xattn_expand(const uint16_t* kv, uint16_t* Kexp, uint16_t* Vexp, int NB, int Tc, int Tcp, int C, int H,
                               hipStream_t s);
hipError_t launch_pack_lnq_t(const float* Wq, const float* gamma, uint16_t* out, int C, float scale, hipStream_t s);
hipError_t launch_xattn_rowstats(const uint16_t* G, const uint16_t* Kexp, const float* bq, float scale, int C, long rows, float* cs,
                                 float* bb, hipStream_t s);
// [Wp.W2 | Wp] operand [C][F + C] and bias Wp.b2 + bp: FeedForward's second Linear (W2 [C][F], b2) merged with proj_out (Wp, bp)
hipError_t launch_pack_ffproj(const float* Wp, const float* bp, const float* W2, const float* b2, uint16_t* wout, float* bout,
                              int C, int F, hipStream_t s);
hipError_t launch_pack_geglu(const float* w, const float* b, uint16_t* wout, float* bout, int half_rows, int K,
                             hipStream_t s);

// ---- sampler elementwise ops on fp32 latents -------------------------------------------------
// e = e_u + scale * (e_c - e_u) for e2 = [e_u ; e_c] (each n elements)
hipError_t launch_bcast_rows(const float* src, float* dst, int rows, int n, hipStream_t s);
hipError_t launch_pack_latent_bcast(const float* x, uint16_t* out, int B, int C, int HW, int cpad, int rep, const float* src,
                                    float* dst, int rows, int n, hipStream_t s);
hipError_t launch_cfg_combine(const float* e2, float* e, long n, float scale, hipStream_t s);
// out = sum_i coef[i] * in[i]   (up to 4 terms; out may alias any input)
hipError_t launch_lincomb(float* out, const float* const* in, const float* coef, int nterms, long n, hipStream_t s);
// DDIM update (ddim.py:258-272): pred_x0 = (x - s1m*e)/sqrt(a_t); x_prev = sqrt(a_prev)*pred_x0 + dir*e + sigma*noise
// inpainting blend: out = (a x0 + b noise) * mask + (1 - mask) * img; mask [B][mask_c][H][W], mask_c = 1 or C
hipError_t launch_q_sample_blend(const float* img, const float* x0, const float* noise, const float* mask, float* out, long n,
                                 long chw, long hw, int mask_c, float a, float b, hipStream_t s);
hipError_t launch_ddim_update(const float* x, const float* e, const float* noise, float* x_prev, float* pred_x0,
                              long n, float sqrt_at, float s1m, float sqrt_aprev, float dir_coef, float sigma,
                              hipStream_t s);
// mean over HW of NHWC fp32 [N][HW][C] -> [N][C]   (classifier head avg-pool)
hipError_t launch_avgpool(const float* x, float* out, int N, int HW, int C, hipStream_t s);

// ---- input-gradient kernels (alignment classifier VJP, csrc/backward.hip) --------------------------------------
hipError_t launch_groupnorm_bwd(const float* x, int ld, int N, int HW, int C, const float* gamma, const float* beta,
                                float eps, int silu, const float* dy, int lddy, const float* addend, int ldadd,
                                float* dx, int lddx, uint16_t* dx_b16, hipStream_t s);
hipError_t launch_layernorm_bwd(const float* x, int rows, int C, const float* gamma, float eps, const float* dy,
                                const float* addend, float* dx, uint16_t* dx_b16, hipStream_t s);
hipError_t launch_geglu_fwd(const uint16_t* u, uint16_t* y, long rows, int H, hipStream_t s);
hipError_t launch_geglu_bwd(const uint16_t* u, const float* dy, uint16_t* du, long rows, int H, hipStream_t s);
// dQ always; dK/dV only when dK != nullptr (cross-attention needs dQ only: the context is a constant)
// Maps that fit no LDS-resident form (more than ~256 tokens: attention at the full resolution, wider latents) take a tiled kernel
// pair that hands the softmax row statistics from the dQ pass to the dK / dV pass through `ws`
// (attention_bwd_ws_floats(...) fp32 values; 0 = a resident form runs, or only dQ is wanted: ws may be null).
size_t attention_bwd_ws_floats(int N, int heads, int D, int Tq, int Tk, int lddk, int lddv, bool want_dkv);
hipError_t launch_attention_bwd(const uint16_t* Q, int ldq, const uint16_t* K, int ldk, const uint16_t* Vt, int ldvt,
                                const float* dO, int lddo, uint16_t* dQ, int lddq, uint16_t* dK, int lddk, uint16_t* dV,
                                int lddv, int N, int heads, int D, int Tq, int Tk, float scale, float* ws, hipStream_t s);
hipError_t launch_cls_head_bwd(const float* prob, const float* w, float* dh, uint16_t* dh_b16, int N, int HW, int C, int Cp,
                               hipStream_t s);
// Linear weight [O][I] fp32 -> transposed bf16 written at out[i*ldo + off + o]  (ldo >= off + O)
hipError_t launch_pack_linear_t(const float* w, uint16_t* out, int O, int I, int ldo, int off, hipStream_t s);
hipError_t launch_pack_conv_bwd(const float* w, uint16_t* out, int O, int I, int Opad, hipStream_t s);

// ---- CAVP video encoder data movement (csrc/cavp.hip) ----------------------------------------------------------
// fp32 NCHW frames [F][3][H][W] -> im2col rows [F*OH*OW][KP] of the (1,7,7) stride-2 pad-3 stem (k = (ky*7+kx)*3 + c)
hipError_t launch_stem_im2col(const float* x, uint16_t* out, int F, int H, int W, int OH, int OW, int KP, hipStream_t s);
hipError_t launch_maxpool3x3s2(const uint16_t* x, uint16_t* out, int F, int H, int W, int OH, int OW, int C, hipStream_t s);
hipError_t launch_subsample2(const uint16_t* x, uint16_t* out, int F, int H, int W, int C, hipStream_t s);
// [F][HW][C] -> [F][HW][3C] = (x[t-1] | x[t] | x[t+1]) with zeros outside each clip of T frames
hipError_t launch_tcat3(const uint16_t* x, uint16_t* out, int F, int T, int HW, int C, hipStream_t s);
// Conv3d weight with eval BatchNorm folded: operand [O][KP] (k = tap*I + i) + fp32 bias [O]
hipError_t launch_pack_conv3d_bn(const float* w, const float* gamma, const float* beta, const float* mean,
                                 const float* var, float eps, uint16_t* out, float* bias, int O, int I, int KT, int KH,
                                 int KW, int KP, hipStream_t s);
// MaxPool1d(k) over the frame axis of [B][T][C] -> [B][T/k][C] (CAVP contrastive head, cavp_model.py:31, 58-59)
hipError_t launch_maxpool_time(const float* x, float* out, int B, int T, int C, int k, hipStream_t s);
hipError_t launch_l2norm_rows(float* x, int rows, int C, hipStream_t s);

// Frame pre-processing in front of the CAVP encoder (Extract_CAVP_Features.forward, demo_util.py:100-104, 150-151):
// frames uint8 [T][H][W][3] RGB -> float [T][3][OH][OW] in [0,1] = torchvision Resize((OH,OW)) on a PIL image
// (PIL.Image.resize BILINEAR, antialiased, 8-bit fixed point) + ToTensor(), bit-identical to Pillow.
// bounds_* [out][2] = (first input index, taps), coef_* int32 [out][ksize] 22-bit fixed point; tmp uint8 [T][H][OW][3].
hipError_t launch_frames_to_tensor(const uint8_t* frames, float* out, uint8_t* tmp, int T, int H, int W, int OH, int OW,
                                   const int* bounds_w, const int* coef_w, int ksize_w, const int* bounds_h,
                                   const int* coef_h, int ksize_h, hipStream_t s);

// ---- mel -> waveform (csrc/vocoder.hip; inverse_op of inference/demo_util.py:196-211) ---------------------------------
// NNLS inversion of the mel filterbank (librosa mel_to_stft, power 1) by FISTA: mel [B][NM][T] normalised log-mel ->
// S [B][T][513] linear magnitude.  A [NM][513], At [513][NM], Pt [NM][513] = pinv(A)^T, inv_L = 1 / sigma_max(A)^2.
hipError_t launch_mel_to_stft(const float* mel, int B, int NM, int T, const float* A, const float* At, const float* Pt,
                              float inv_L, int iters, float* S, hipStream_t s);
// Fast Griffin-Lim (librosa griffinlim: n_fft 1024, hop 256, hann, centre): S [B][T][513], phase0 [B][513][T] in [0,1)
// (the rng.rand draw of init="random"), tw[512] = exp(-2 pi i k/1024), window[1024], wss[1024 + 256 (T-1)] window
// sum-square; workspaces angles / reb0 / reb1 complex [B][T][513], frames [B][T][1024]; y [B][256 (T-1)] = the waveform.
hipError_t launch_griffinlim(const float* S, const float* phase0, int B, int T, int n_iter, float momentum, const float2* tw,
                             const float* window, const float* wss, float2* angles, float2* reb0, float2* reb1, float* frames,
                             float* y, hipStream_t s);
