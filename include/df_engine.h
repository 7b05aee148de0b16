/* df_engine.h -- C ABI of libdfengine.so, the MI355X (gfx950) engine behind the Diff-Foley
 * Stage-2 sampling path.
 *
 * The reference (luosiallen/Diff-Foley) has NO FFI/plugin boundary: the hot path is a tree of
 * torch.nn modules driven from Python.  Each entry point below replaces one Python-level call
 * of the reference; the reference-side binding a maintainer would add is the ctypes stub shown
 * in INTEGRATION.md (and implemented in diff_foley_amd/engine.py).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; df_last_error() gives the message
 *     (thread-local); nothing throws across the ABI.
 *   - "dev" pointers are device pointers (torch tensor.data_ptr()); the caller owns every I/O buffer,
 *     the library owns packed weights and workspaces for the lifetime of the ctx.
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*); no internal
 *     synchronisation except in df_load_tensor / df_finalize / df_autotune.
 *   - latents are NCHW fp32 exactly as the reference passes them; internal layout is NHWC.
 *   - one ctx per device per process; a ctx is not re-entrant.
 */
#ifndef DF_ENGINE_H
#define DF_ENGINE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct df_ctx df_ctx;

/* UNetModel / Classifier_Backbone hyper-parameters
 * (diff_foley/modules/diffusionmodules/openai_unetmodel.py:443-468, inference/config/Stage2_LDM.yaml:21-36,
 *  inference/config/Double_Guidance_Classifier.yaml:35-50). */
typedef struct {
  int in_channels, out_channels, model_channels, num_res_blocks;
  int channel_mult[8];
  int n_mult;
  int attention_resolutions[8];
  int n_attn;
  int num_heads;
  int context_dim;
} df_unet_config;

/* AutoencoderKL decoder ddconfig (Stage2_LDM.yaml:38-59) + LatentDiffusion.scale_factor (:17). */
typedef struct {
  int z_channels, embed_dim, ch, num_res_blocks, out_ch;
  int ch_mult[8];
  int n_mult;
  float scale_factor;
} df_vae_config;

/* Video_Feat_Encoder_Posembed (Stage2_LDM.yaml:62-67). */
typedef struct {
  int origin_dim, embed_dim, seq_len;
} df_cond_config;

/* CAVP video encoder = ResNet3dSlowOnly(depth 50) + video_project_head (inference/model/cavp_model.py:21-29,
 * inference/model/cavp_modules.py:1233-1268): blocks per stage (3,4,6,3), stem width 64, feature width embed_dim. */
typedef struct {
  int stage_blocks[4];
  int base_channels;
  int embed_dim;
} df_cavp_config;

/* ---- lifetime ------------------------------------------------------------------------------ */
int df_create(int device, df_ctx** out);
void df_destroy(df_ctx* ctx);
const char* df_last_error(void);
int df_abi_version(void);
/* MFMA operand type this build was compiled for: "bf16" (libdfengine.so) or "f16" (libdfengine_f16.so, built from
 * the same sources with -DDF_OPERAND_F16).  2-byte operand buffers passed to the df_test_* entry points use it. */
const char* df_operand_dtype(void);

/* ---- model definition: replaces instantiate_from_config(config.model) + load_state_dict()
 *      (inference/diff_foley_inference.ipynb:80-95; diff_foley/util.py:176-191).
 * `name` is the reference state_dict key (e.g. "model.diffusion_model.input_blocks.1.0.in_layers.2.weight");
 * classifier tensors are loaded under the prefix "classifier." + their own key ("classifier.model....").
 * `host` is fp32, C-contiguous; it is copied before the call returns. */
int df_config_unet(df_ctx* ctx, const df_unet_config* cfg);
int df_config_vae(df_ctx* ctx, const df_vae_config* cfg);
int df_config_cond(df_ctx* ctx, const df_cond_config* cfg);
int df_config_classifier(df_ctx* ctx, const df_unet_config* cfg);
/* Tensors of the CAVP video branch are loaded under "cavp." + the CAVP_Inference state_dict key
 * ("cavp.video_encoder.conv1.conv.weight", "cavp.video_encoder.layer1.0.conv1.bn.running_var",
 * "cavp.video_project_head.weight", ...). */
int df_config_cavp(df_ctx* ctx, const df_cavp_config* cfg);
int df_load_tensor(df_ctx* ctx, const char* name, const float* host, const int64_t* shape, int ndim);
/* Same, but the fp32 source already lives on the device (used after an RCCL weight broadcast). */
int df_load_tensor_dev(df_ctx* ctx, const char* name, const float* dev, const int64_t* shape, int ndim);
/* Checks every tensor the configured modules need is present and re-packs to bf16 MFMA layouts. */
int df_finalize(df_ctx* ctx);
/* Optional: time the candidate tile shapes of every GEMM of the plans built so far and keep the fastest. */
int df_autotune(df_ctx* ctx, int enable);

/* ---- LatentDiffusion.get_learned_conditioning (ddpm.py:568-579 -> video_feat_encoder.py:12-18)
 * feats [B][T][origin_dim] fp32 -> out [B][T][embed_dim] fp32 */
int df_cond_encode(df_ctx* ctx, const float* feats_dev, float* out_dev, int B, int T, void* stream);

/* ---- CAVP_Inference.encode_video(video, normalize, pool=False) (inference/model/cavp_model.py:47-65), as called by
 * Extract_CAVP_Features.forward (inference/demo_util.py:150-167): video [B][T][3][H][W] fp32 RGB in [0,1] (H, W
 * multiples of 32; the reference uses 224) -> out [B][T][embed_dim] fp32, rows L2-normalised when normalize != 0.
 * Clips are independent; the temporal (3,1,1) convolutions zero-pad inside each clip of T frames. */
int df_cavp_encode(df_ctx* ctx, const float* video_dev, float* out_dev, int B, int T, int H, int W, int normalize,
                   void* stream);

/* The pooled form, encode_video(..., pool=True) (cavp_model.py:58-59: nn.MaxPool1d(kernel_size=16) over the frame axis of the
 * projected features, then the optional F.normalize): feat [B][T][C] fp32 (df_cavp_encode with normalize = 0) -> out
 * [B][T / kernel][C], rows L2-normalised when normalize != 0. */
int df_cavp_pool(const float* feat_dev, float* out_dev, int B, int T, int C, int kernel, int normalize, void* stream);

/* ---- UNetModel.forward (openai_unetmodel.py:710-742) through LatentDiffusion.apply_model (ddpm.py:925-1026).
 * The cross-attention context is step-invariant, so it is set once per sample() call: everything of the 16
 * SpatialTransformers' cross-attention that depends on the context only is computed here, not per step -- the K/V
 * projections and, for T <= 32, their products with the (LayerNorm-folded) query and output projections, so that a step
 * runs cross-attention as two GEMMs against per-sample operands.  context [N][T][context_dim] fp32. */
int df_unet_set_context(df_ctx* ctx, const float* context_dev, int N, int T, void* stream);
/* x [N][C][H][W] fp32, t [N] fp32 (integer or fractional timesteps), eps_out [N][C][H][W] fp32. */
int df_unet_forward(df_ctx* ctx, const float* x_dev, const float* t_dev, float* eps_out_dev, int N, int H, int W,
                    void* stream);
/* Classifier-free-guidance step of p_sample_ddim (ddim.py:241-245): runs the UNet on cat([x,x]) against the
 * 2B-row context set before ([uncond ; cond]) and returns e_u + scale*(e_c - e_u).  x, t, eps: B rows.  The ops in front of
 * the first cross-attention see identical rows in both halves and run on one half only (DF_NO_CFGDEDUP=1 disables). */
int df_unet_forward_cfg(df_ctx* ctx, const float* x_dev, const float* t_dev, float* eps_out_dev, int B, int H, int W,
                        float guidance_scale, void* stream);
/* Time embedding of a whole sample() call, hoisted out of the step loop like the context (timestep_embedding util.py:151-171 ->
 * time_embed openai_unetmodel.py:506-511,724 -> every ResBlock's emb_layers :262: all of it depends on t only, and a sampler
 * knows its S timesteps before the loop -- ddim.py:193-201, plms.py:127-135, dpm_solver.py:1071-1079).  t_host[S]: the
 * timesteps (host memory; integer or fractional), each used for every sample of the batch as the reference samplers do
 * (ddim.py:217).  N, H, W, cfg name the plan (cfg != 0: the CFG plan of df_unet_forward_cfg with B = N).  Call after
 * df_unet_set_context.  The *_ts entry points then take the INDEX of the step's timestep in that table instead of t_dev and
 * replace the four time-embedding launches of a step by one table look-up; results are bit-identical to the t_dev forms.
 * Announcing the timesteps the plan's table already holds returns at once (a service's sample() calls repeat the same 25 / 50). */
int df_unet_set_timesteps(df_ctx* ctx, const float* t_host, int S, int N, int H, int W, int cfg, void* stream);
int df_unet_forward_ts(df_ctx* ctx, const float* x_dev, int ts_index, float* eps_out_dev, int N, int H, int W, void* stream);
int df_unet_forward_cfg_ts(df_ctx* ctx, const float* x_dev, int ts_index, float* eps_out_dev, int B, int H, int W,
                           float guidance_scale, void* stream);

/* ---- LatentDiffusion.decode_first_stage (ddpm.py:739-797 -> autoencoder.py:330-333 -> model.py:630-663)
 * z [B][z_channels][H][W] fp32 -> out [B][out_ch][H*2^(n_mult-1)][W*2^(n_mult-1)] fp32.  Any B: batches whose widest
 * activation would exceed the 2 GiB operand addressing (B > 16 at the full decoder) run as slices inside this call. */
int df_vae_decode(df_ctx* ctx, const float* z_dev, float* out_dev, int B, int H, int W, void* stream);

/* ---- Alignment classifier forward (alignment_classifier.py:269-271 -> alignment_backbone.py:656-686)
 * x [B][C][H][W], t [B], video_feat [B][T][context_dim] (raw CAVP features) -> prob [B][out_channels] */
int df_classifier_forward(df_ctx* ctx, const float* x_dev, const float* t_dev, const float* feat_dev, float* prob_dev,
                          int B, int H, int W, int T, void* stream);

/* Classifier guidance gradient (cal_classifier_loglikelihood_grad, ddim.py:333-341; cond_grad_fn_classifier,
 * dpm_solver.py:1340-1349):  grad = d sum_b log p_b / d x  (UNSCALED), shape of x; prob (optional, may be NULL) [B][1].
 * Forward and hand-written backward-data pass run natively (no autograd). */
int df_classifier_grad(df_ctx* ctx, const float* x_dev, const float* t_dev, const float* feat_dev, float* prob_dev,
                       float* grad_dev, int B, int H, int W, int T, void* stream);

/* The same call for a guidance loop, which hands the classifier the SAME video features at every step (ddim.py:374-380 passes
 * origin_cond unchanged through all S steps; dpm_solver.py:1377-1393 closes over it): feat_token != 0 names the CONTENTS of
 * feat_dev.  While consecutive calls on a plan carry the same token, the feature-only launches (cast + cross-attention K / V^T
 * of every transformer block: 7 of the call's launches at the Stage-2 classifier) are skipped and feat_dev is not read.  The
 * caller changes the token whenever the features change; token 0 = no reuse (df_classifier_grad).  The plan holds the token:
 * a plan that was rebuilt (other shape, reloaded weights) recomputes whatever token it is handed first. */
int df_classifier_grad_cached(df_ctx* ctx, const float* x_dev, const float* t_dev, const float* feat_dev, float* prob_dev,
                              float* grad_dev, int B, int H, int W, int T, uint64_t feat_token, void* stream);

/* ---- video frame pre-processing in front of the CAVP encoder (Extract_CAVP_Features.forward, inference/demo_util.py:
 * 100-104, 150-151): per frame torchvision Resize((OH, OW)) on a PIL image (= PIL.Image.resize BILINEAR: antialiased,
 * 8-bit fixed point, horizontal pass then vertical pass -- vertical pass first on frames with H > 100 W and OH < H, as Pillow does)
 * + ToTensor().  frames uint8 [T][H][W][3] RGB ->
 * out fp32 [T][3][OH][OW] in [0, 1], bit-identical to Pillow.  bounds_* int32 [out][2] = (first input index, taps),
 * coef_* int32 [out][ksize] 22-bit fixed-point filter weights (computed in double precision on the host:
 * diff_foley_amd/video.py); tmp uint8 scratch for the first pass's result: [T][H][OW][3], or [T][OH][W][3] when H > 100 W and OH < H.
 * All pointers device memory. */
int df_frames_to_tensor(const uint8_t* frames_dev, float* out_dev, uint8_t* tmp_dev, int T, int H, int W, int OH, int OW,
                        const int32_t* bounds_w_dev, const int32_t* coef_w_dev, int ksize_w, const int32_t* bounds_h_dev,
                        const int32_t* coef_h_dev, int ksize_h, void* stream);

/* ---- mel -> waveform (inverse_op, inference/demo_util.py:196-211; algorithms of librosa 0.8.0, the
 * reference's pinned dependency).  df_mel_to_stft: undo the log-mel normalisation and invert the mel filterbank by
 * non-negative least squares (librosa.feature.inverse.mel_to_stft, power 1): mel [B][n_mels][T] -> S [B][T][513].
 * A [n_mels][513] = librosa.filters.mel(22050, 1024, n_mels, 125, 7600), At its transpose, Pt [n_mels][513] = pinv(A)^T
 * (the clipped least-squares start of librosa.util.nnls), inv_L = 1 / sigma_max(A)^2, iters FISTA iterations.
 * df_griffinlim: librosa.griffinlim(S, hop_length=256): 32 fast Griffin-Lim iterations, momentum 0.99; phase0
 * [B][513][T] uniform in [0,1) is the random initial phase; twiddles complex [512] exp(-2 pi i k/1024), window [1024]
 * periodic hann, wss [1024 + 256 (T-1)] window sum-square; workspaces: angles / reb0 / reb1 complex [B][T][513], frames
 * [B][T][1024]; wav [B][256 (T-1)].  All pointers device memory, all launches asynchronous on `stream`. */
int df_mel_to_stft(const float* mel_dev, int B, int n_mels, int T, const float* A_dev, const float* At_dev,
                   const float* Pt_dev, float inv_L, int iters, float* S_dev, void* stream);
int df_griffinlim(const float* S_dev, const float* phase0_dev, int B, int T, int n_iter, float momentum,
                  const float* twiddles_dev, const float* window_dev, const float* wss_dev, float* angles_dev,
                  float* reb0_dev, float* reb1_dev, float* frames_dev, float* wav_dev, void* stream);

/* ---- packed-operand blob (multi-GPU weight distribution; replaces SURVEY's df_bcast_weights: the RCCL
 * communicator belongs to torch.distributed, so the library exports / imports and the host side broadcasts).
 * Root rank: load tensors, df_finalize, df_prepack (builds every operand packing the UNet CFG-batch 2B / VAE / cond plans
 * of this shape use), df_packed_size, df_export_packed.  Other ranks: df_create, df_config_*, df_import_packed(manifest,
 * blob), df_finalize -- no fp32 master copies, no re-packing.  The manifest (host bytes) lists every tensor's shape, the
 * small fp32 tensors that plans read directly (biases, norm parameters, pos_emb) and every packed operand with its
 * offset into the blob (device bytes, 256-B aligned segments).  Operand type (bf16 / fp16 build) is checked on import. */
int df_prepack(df_ctx* ctx, int B, int H, int W, int T);
int df_packed_size(df_ctx* ctx, size_t* manifest_bytes, size_t* blob_bytes);
int df_export_packed(df_ctx* ctx, void* manifest_host, void* blob_dev, void* stream);
int df_import_packed(df_ctx* ctx, const void* manifest_host, size_t manifest_bytes, const void* blob_dev,
                     size_t blob_bytes, void* stream);

/* ---- sampler arithmetic on fp32 latents (n = number of elements) -------------------------------
 * e = e_u + scale*(e_c - e_u), e2 = [e_u ; e_c]                     (ddim.py:245, dpm_solver.py:1386) */
int df_cfg_combine(const float* e2_dev, float* e_dev, int64_t n, float scale, void* stream);
/* out = sum_i coef[i]*in[i], 1..4 terms; out may alias an input  (DPM-Solver++ updates dpm_solver.py:504-549,
 * 755-810; PLMS multistep plms.py:219-232; classifier guidance ddim.py:380) */
int df_lincomb(float* out_dev, const float* const* in_dev, const float* coef, int nterms, int64_t n, void* stream);
/* Inpainting blend of the samplers (reference ddim.py:206-209, plms.py:147-150, ddpm.py:1239-1241 with q_sample ddpm.py:279-282):
 * out = (sqrt_acp * x0 + sqrt_one_minus_acp * noise) * mask + (1 - mask) * img.  All tensors fp32 NCHW [B][C][H][W] (n elements,
 * chw per sample, hw per channel); mask is [B][mask_c][H][W] with mask_c = 1 (broadcast over channels) or C.  out != img. */
int df_q_sample_blend(const float* img_dev, const float* x0_dev, const float* noise_dev, const float* mask_dev, float* out_dev,
                      int64_t n, int64_t chw, int64_t hw, int mask_c, float sqrt_acp, float sqrt_one_minus_acp, void* stream);
/* DDIM update (ddim.py:258-272).  noise may be NULL (eta = 0). */
int df_ddim_update(const float* x_dev, const float* e_dev, const float* noise_dev, float* x_prev_dev,
                   float* pred_x0_dev, int64_t n, float a_t, float a_prev, float sigma_t, float sqrt_one_minus_at,
                   void* stream);

/* ---- introspection for bench.py / tests --------------------------------------------------------- */
/* Plans (static launch lists + workspaces, one per (network, batch, latent, context) shape) currently cached and the bytes
 * of HBM they own.  The cache is bounded: beyond DF_MAX_PLANS (env, default 32) the least recently used plan is dropped. */
int df_plan_count(df_ctx* ctx, int64_t* n_plans, int64_t* workspace_bytes);
/* Number of kernel launches of the last UNet plan executed and its algorithmic GEMM FLOPs. */
int df_unet_plan_stats(df_ctx* ctx, int64_t* n_launches, double* gemm_flops, double* weight_bytes);
/* Per-op-family HIP-event timing of everything executed between begin and end (events are recorded on the
 * stream the kernels are launched on).  Families: 0 MFMA implicit-GEMM (conv3x3/1x1/linear/batched, incl. split-K
 * reduce), 1 fused attention, 2 GroupNorm, 3 LayerNorm, 4 other (packing, embeddings, sampler arithmetic).
 * ms_by_family / count_by_family: arrays of 5. */
int df_profile_begin(df_ctx* ctx);
int df_profile_end(df_ctx* ctx, double* ms_by_family, int64_t* count_by_family);
/* CSV (tag,M,N,K,taps,stride,ups,batch,tile,splitk,ms) of every op of the region last closed by df_profile_end. */
int df_profile_dump(df_ctx* ctx, const char* path);
/* The autotuner's choices as text ("key tile splitk gm" per line; process-wide).  Export: *n = bytes needed, at most cap are
 * written to buf (buf may be NULL to query the size).  Import merges the lines into this process's cache: plans built with
 * df_autotune(1) whose GEMMs are all present are configured from it without a trial launch.  parallel.broadcast_packed_model
 * ships rank 0's text with the packed blob: every rank runs rank 0's tiles (identical fp32 summation order). */
int df_tune_cache_export(char* buf, int64_t cap, int64_t* n);
int df_tune_cache_import(const char* text, int64_t n);
/* Debug: while enabled, every op of every plan is followed by a 64-bit (order-independent, integer) checksum over ALL
 * workspace bytes of its plan; the sequence of checksums of two runs on identical inputs must be identical, and the first
 * index that differs names the launch that was not reproducible (tools/chk_probe.py).  capacity = checksum slots. */
int df_debug_checksums(df_ctx* ctx, int enable, int64_t capacity);
int df_debug_checksums_read(df_ctx* ctx, uint64_t* out, int64_t cap, int64_t* n);
int df_debug_checksum_label(df_ctx* ctx, int64_t index, char* buf, int64_t len);
/* Debug: while enabled, every op of every plan is followed by a count of the operand-type values it stored that sit at the
 * saturation point of the operand format -- fp16 build: |v| == 65504, where every fp32 -> fp16 conversion of the kernels
 * clamps instead of overflowing (csrc/common.h op_clamp); bf16 build: non-finite values.  Covers the operand-type outputs of
 * every GEMM epilogue (C, the aux copy, the transposed V) and of the norm / cast / attention ops.  A non-zero count means the
 * fp16 build has lost information in that op (a trained checkpoint's activations left the fp16 range): use the bf16 build.
 * The reference computes in fp32 and has no such limit (openai_unetmodel.py:24-28: convert_module_to_f16 is a stub). */
int df_debug_saturations(df_ctx* ctx, int enable, int64_t capacity);
int df_debug_saturations_read(df_ctx* ctx, uint64_t* out, int64_t cap, int64_t* n);
/* Error budget of the bf16 build (tools/error_budget.py; no reference counterpart): in the fp16 build, re-round the operand-type outputs of
 * every op whose tag starts with one of the comma-separated prefixes ("*" = all, "" = off) to bf16's 8 significant bits right behind
 * the op.  The bf16 build accepts the call and changes nothing. */
int df_debug_requant(df_ctx* ctx, const char* tag_prefixes);
int df_debug_saturation_label(df_ctx* ctx, int64_t index, char* buf, int64_t len);
/* Run ONE op family in isolation for unit tests (see tests/test_kernels_gpu.py). */
int df_test_scratch_read(void* host, int64_t bytes);     /* the shared scratch of the test entry points (debug stamps) */
int df_test_geglu(const uint16_t* A_dev, const uint16_t* W_dev, const void* stats_dev, const float* cs_dev, const float* bias_dev,
                  uint16_t* out_dev, int M, int K, int N1, int tile, int dbg, void* stream);
int df_test_conv3x3_fewout(const uint16_t* A_nhwc_dev, const uint16_t* W_okki_dev, const float* bias_dev, float* out_nchw_dev, int NB,
                           int H, int W, int Cin, int Cout /* <= 4 */, void* stream);
int df_test_gemm_epi(const uint16_t* A_dev, const uint16_t* W_dev, const float* bias_dev, const float* res_dev, void* C_dev,
                     int M, int N, int K, int act /*0 none, 1 SiLU, 2 ReLU*/, int out_operand, int tile, int splitk,
                     void* stream);
int df_test_gemm_dual(const uint16_t* A_dev, const uint16_t* A2_dev, const uint16_t* W_dev, float* C_dev, int M, int N, int K1,
                      int K2, int tile, int splitk, void* stream);
int df_test_gemm(const uint16_t* A_dev, const uint16_t* W_dev, float* C_dev, int M, int N, int K, int tile, int splitk,
                 void* stream);
/* SpatialTransformer GEMM pair: producer (fp32 t0 + operand copy + per-row partial statistics from the epilogue) and
 * LayerNorm-folded consumer (attention_openai.py:211-215 pre-norms never run as kernels).  mode 0 plain fp32 out,
 * 1 GEGLU, 2 fused QKV with the V third stored transposed per sample of T rows. */
int df_test_ln_chain(const uint16_t* A0_dev, const uint16_t* W0_dev, const float* b0_dev, const float* res_in_dev,
                     const float* gamma_dev, const float* beta_dev, const float* W1_dev, const float* b1_dev,
                     float* t0_dev, void* y_dev, uint16_t* vt_dev, int M, int C, int N1, int mode, int T, int ldvt,
                     int tile0, int sk0, int tile1, int sk1, void* stream);
/* Small-M weight-streaming linear (time-embedding path): lds_variant 1 = activations staged in LDS (tvals != NULL:
 * the activations are the sinusoidal embedding of tvals[m % t_B]); 0 = the register variant. */
int df_test_linear_rows(const float* a_dev, int lda, const float* tvals_dev, int t_B, const uint16_t* W_dev,
                        const float* bias_dev, float* out_dev, int ldo, int M, int N, int K, int act, int lds_variant,
                        void* stream);
/* ONE block of the loaded UNet in isolation (checked against the reference's per-block tensors, golden G3): kind 0
 * ResBlock (openai_unetmodel.py:255-275; semb = SiLU(time_embed(t))), 1 SpatialTransformer (attention_openai.py:250-261),
 * 2 Downsample, 3 Upsample.  x [N*H*W][Cin] -> out [N*OH*OW][Cout], NHWC fp32; prefix e.g. "input_blocks.1.0". */
int df_test_unet_block(df_ctx* ctx, const char* prefix, int kind, const float* x_dev, const float* semb_dev,
                       const float* context_dev, float* out_dev, int N, int H, int W, int Cin, int Cout, int T, void* stream);
int df_test_conv3x3(const uint16_t* A_dev, const uint16_t* W_dev, const float* bias_dev, float* C_dev, int NB, int H,
                    int W, int Cin, int Cout, int stride, int ups, int tile, int splitk, void* stream);
/* conv3x3 (stride 1) with a folded 1x1 skip connection: C = conv3x3(A; W[:, :9 Cin]) + A2 . W[:, 9 Cin:]^T + bias, one implicit GEMM
 * with K = 9 Cin + Cin2 (A [NB*H*Wd][Cin], A2 [NB*H*Wd][Cin2], W [Cout][9 Cin + Cin2] operand type; Cin, Cin2 multiples of 64). */
int df_test_conv3x3_skip(const uint16_t* A_dev, const uint16_t* A2_dev, const uint16_t* W_dev, const float* bias_dev, float* C_dev,
                         int NB, int H, int W, int Cin, int Cin2, int Cout, int tile, int splitk, void* stream);
/* nearest-x2 upsample + conv3x3 in the phase-decomposed form (four 2x2-tap convs on the input-resolution map, per-phase
 * weights = sums of the 3x3 taps): A [NB*H*Wd][Cin] operand type, W_oihw fp32 [Cout][Cin][3][3], w4_scratch 16*Cout*Cin
 * operand-type elements, C fp32 [NB*2H*2Wd][Cout]. */
int df_test_conv3x3_ups4(const uint16_t* A_dev, const float* W_oihw_dev, const float* bias_dev, float* C_dev, uint16_t* w4_scratch_dev,
                         int NB, int H, int Wd, int Cin, int Cout, int tile, int splitk, void* stream);
int df_test_groupnorm(const float* x_dev, int ld, int N, int HW, int C, const float* gamma, const float* beta, float eps,
                      int silu, uint16_t* out_dev, void* stream);
/* GroupNorm whose input is the not-yet-reduced output of its own split-K producer: channels [0, c_own) of x are summed from
 * `nslab` fp32 slabs [N*HW][c_own] (consecutive), + bias[c_own] + res[N*HW][ldr] (either may be NULL), written back to x and
 * normalised together with channels [c_own, C) read from x. */
int df_test_groupnorm_own_slabs(float* x_dev, int ld, int N, int HW, int C, const float* gamma, const float* beta, float eps,
                                int silu, uint16_t* out_dev, const float* slabs_dev, int nslab, int c_own,
                                const float* bias_dev, const float* res_dev, int ldr, void* stream);
int df_test_layernorm(const float* x_dev, int rows, int C, const float* gamma, const float* beta, uint16_t* out_dev,
                      void* stream);
int df_test_attention(const uint16_t* Q, int ldq, const uint16_t* K, int ldk, const uint16_t* Vt, int ldvt, uint16_t* O,
                      int ldo, int N, int heads, int D, int Tq, int Tk, float scale, void* stream);
/* Device-peak microbenchmarks (tools/peaks.py; SURVEY.md 8d "peaks measured on the box").  kind 0: MFMA issue peak
 * (n = iterations per wavefront of 4 independent v_mfma_f32_32x32x16_bf16; grid blocks x 256 threads); kind 1:
 * streaming copy of n bytes; kind 2: streaming read of n bytes. */
int df_test_peak(int kind, const void* src_dev, void* dst_dev, size_t n, int blocks, void* stream);
/* L2 -> CU fill-rate probes: kind 3 LDS-DMA, 4 global_load_dwordx4 -> VGPR, 5 global_load + ds_write_b128.  Every block
 * re-reads its span-byte window n times; stride_blk bytes between the windows of consecutive blocks. */
int df_test_fill(int kind, const void* src_dev, void* dst_dev, size_t span, size_t stride_blk, int n, int blocks,
                 void* stream);


#ifdef __cplusplus
}
#endif
#endif /* DF_ENGINE_H */
