// Implicit-GEMM descriptor shared by host launcher and device kernels.
//
//   C[m, n] = epilogue( alpha * sum_k A(m, k) * W[n, k] )
//
// A is a bf16 NHWC activation tensor [rows][lda]; for taps == 9 the k index runs over
// (ky, kx, cin) of a 3x3 window (zero padding 1, stride 1|2, optional nearest x2 upsample of
// the input), i.e. the convolution is evaluated as a GEMM without materialising im2col.
// W is bf16 [N][K] (K contiguous) -- PyTorch Linear layout, conv weights re-packed to
// [Cout][ky][kx][Cin] at load time.  Accumulation is fp32 on the MFMA units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

struct GemmParams {
  // operands
  const uint16_t* A; long a_bs; int lda;
  const uint16_t* W; long w_bs;
  unsigned a_bytes, w_bytes;   // bytes addressable from A / W of ONE batch slice (buffer-load bounds, < 2 GiB)
  int M, N, K;
  // implicit-conv geometry
  // ResBlock skip connection folded into conv2 (openai_unetmodel.py:234-241, 275: skip(x) + h): a TENTH K range of
  // Cin2 channels read from a second operand tensor at the centre pixel, K = 9*Cin + Cin2, W rows = [conv taps | skip]
  const uint16_t* A2; int lda2; unsigned a2_bytes; int Cin2;
  int taps;    // 1 (linear / 1x1) or 9 (3x3, pad 1)
  int Cin;     // channels per tap, K = taps * Cin, Cin % 64 == 0
  int H, Wd;   // stored input spatial size
  int OH, OW;  // output spatial size (M = batch * OH * OW)
  int stride;  // 1 | 2
  int ups;     // 1: conv runs on the nearest-x2 upsampled input
  int zstuff;  // with ups=1: the x2 input is ZERO-stuffed (transposed stride-2 conv, backward of Downsample), not nearest
  int th, tw;  // halo kernels: spatial patch of output pixels owned per block (th*tw divides BM)
  int halo_ring_bytes;   // halo kernels (set by the launcher): LDS bytes of the operand ring; the epilogue row table follows
  // epilogue
  void* C; long c_bs; int ldc; int out_bf16;
  float alpha;
  const float* bias;                 // [N]
  const float* rowbias; int ld_rowbias; int rows_per_sample; int rowbias_mode;  // 1: by sample, 2: by position
  const float* res; long res_bs; int ldr;   // fp32 residual, may alias C
  int relu;                          // max(v, 0) after bias / residual (CAVP encoder ConvModule activation)
  int silu;                          // v * sigmoid(v) after bias (time-embedding MLP layers); LEAN / ANY epilogues
  uint16_t* aux; int ld_aux;         // optional second output: operand-type copy of the stored value, [row][ld_aux]
  int geglu;                         // columns come in (x:32 | gate:32) groups, output width N/2
  // LayerNorm folded into this GEMM (A = raw operand copy of x, W = gamma-scaled weights):
  //   out[m][n] = rstd[m] * (acc[m][n] - mean[m] * ln_cs[n]) + bias[n]      (bias already holds beta.W + b)
  // mean / rstd of row m come from the producer's per-row partials ln_stats[m][0..ln_slots) = (sum, sum of squares)
  // over 64-column slots of the fp32 tensor the operand copy was made from (ln_C columns in total).
  const float2* ln_stats; int ln_slots; int ln_C; float ln_eps;
  const float* ln_cs;                // [N] column sums of the (operand-rounded) gamma-scaled weights
  // The same GEGLU projection in the 320-column packing of the wide tiles (ffn_wide.hip, round 6): weight rows, column sums and
  // folded bias permuted so that every 160-row half of a 320-row tile is [80 x rows | their 80 gate rows].  null = not packed.
  const uint16_t* W_w320; const float* cs_w320; const float* bias_w320;
  // Row-block weights: rows [i*w_rows, (i+1)*w_rows) multiply W + i*w_bs (one weight matrix per SAMPLE in one launch: the
  // cross-attention GEMMs whose "weights" are precomputed from each sample's context).  w_rows % BM == 0.
  int w_rows;
  // Cross-attention score epilogue (with ln_stats): ln_cs and bias are per sample, [M / w_rows][N]; every group of sm_w = 32
  // columns (one head) gets a softmax over its first sm_valid columns (the rest is padding and stores 0); operand-type output.
  int sm_w, sm_valid;
  // per-row partial statistics of the STORED fp32 value (after bias / residual): stats[row][col/64] = (sum, sumsq)
  float2* stats; int stats_slots;
  // columns >= vt_col0 are stored TRANSPOSED per sample into vt[(row/vt_T)*(N-vt_col0) + col-vt_col0][ldvt] at
  // position row%vt_T (attention V^T straight out of the fused QKV projection); vt_col0 % BN == 0 for every tile used
  uint16_t* vt; int vt_col0; int vt_T; int ldvt;
  // Classifier-free-guidance prefix: the two halves of the CFG batch are identical until the first cross-attention, so the
  // ops in front of it run on ONE half (M rows) and the op that feeds the full batch stores every output row twice, at
  // row and row + dup_rows (C / aux / stats alike).  0 = off.  Plain row-major epilogues only (LEAN / PROD / ANY, reduce).
  int dup_rows;
  int no_c_store;   // PROD epilogue: skip the fp32 store of C (nobody reads it); operand copy + row statistics only
  int store_nchw; int hw_out;        // write C as [batch][N][hw_out] instead of [rows][ldc]
  int gm;      // tile walk: each XCD's contiguous tile range runs M-fastest inside row groups of `gm` M-tiles (0 = all rows:
               // plain M-fastest; 1 = N-fastest).  Decides which operand panels an XCD's L2 can share; autotuned in situ.
  int dbg;     // tools only: 1 = every K tile re-reads tile 0 (cache-resident operands; isolates memory latency)
  // split-K
  int splitk; float* partial;
  int defer_reduce;   // split-K only: write the partial slabs and do NOT launch the reduce -- the consumer (GroupNorm) sums them
  // split-K + NCHW store of a CFG batch [uncond ; cond] (UNetModel.out): the reduce launch also forms e_u + cfg_scale (e_c - e_u)
  // and writes THAT, [M / 2 rows][N] as NCHW, to cfg_out (ddim.py:241-245); C is not written.  null = off.
  float* cfg_out; float cfg_scale;
};

#if defined(__HIP_DEVICE_COMPILE__)
// kernel-argument / own-code touch of the GEMM kernels: common.h df_entry_touch
__device__ __forceinline__ DfTouch gemm_kernarg_touch() { return df_entry_touch((int)sizeof(GemmParams)); }
__device__ __forceinline__ void gemm_kernarg_touch_end(const DfTouch& v) { df_entry_touch_end(v); }
#endif

enum GemmTile {
  TILE_128x128 = 0, TILE_128x64 = 1, TILE_64x128 = 2, TILE_64x64 = 3, TILE_32x128 = 4, TILE_COUNT = 5,   // generic
  // conv3x3 stride-1 kernels with an LDS-staged halo tile (BM output pixels = patches of th x tw, BN couts)
  TILE_HALO_128x64 = 5, TILE_HALO_256x64 = 6, TILE_HALO_128x128 = 7,
  // generic kernel with 8 wavefronts (512 threads): high arithmetic intensity per LDS byte, for split-K streaming
  TILE_128x256 = 8, TILE_256x128 = 9,
  // generic kernel, double-buffered (ring depth 2): less LDS -> 2-5 resident blocks per CU, for short-K layers
  TILE_128x128_S = 10, TILE_128x64_S = 11, TILE_64x128_S = 12, TILE_64x64_S = 13, TILE_32x128_S = 14,
  // halo kernels with an 8-deep weight ring: more weight bytes in flight per CU for the weight-streaming layers
  TILE_HALO_128x64_D = 15, TILE_HALO_256x64_D = 16,
  // 192 output pixels (three 4x16 patches) x 64 couts, 4 wavefronts of 96x32: (8192, 320) becomes 43 x 5 = 215 blocks on 256 CUs
  // where the 256x64 tile gives 160 (the model's channel counts are 5 * 2^k: power-of-two tiles leave 3/8 of the CUs idle)
  TILE_HALO_192x64 = 17,
  // PRODUCER-SPECIALISED generic tiles (round 4; gemm_impl.h PS): 4 consumer wavefronts (2 x 2) run the MFMAs, 4 producer
  // wavefronts issue the LDS-DMA requests of the ring -- 8 wavefronts, one of each role per SIMD.  (ids 18-20 were the deep
  // weight-ring tiles of an earlier round-4 experiment: experiments/deep_weight_ring_tiles.md; never built into the product.)
  TILE_PS_256x128 = 18, TILE_PS_128x128 = 19, TILE_PS2_128x128 = 20,      // PS2: 8 producer wavefronts (12 in the block)
  // persistent LayerNorm-folded GEGLU projection (ffn.hip): 2 resident blocks per CU walk a tile queue, one continuous operand
  // stream across tiles, epilogue out of the accumulator registers.  128 x 128 tiles (C <= 640) / 64 x 128 tiles.
  TILE_PGEGLU_128 = 21, TILE_PGEGLU_64 = 22,
  // producer-specialised halo conv tiles: 4 consumer + 4 producer wavefronts.  (8 producers -- 12 wavefronts, 168 VGPRs each --
  // spill in the tap loop: 32 us against 21.8 us on the 320 -> 320 conv at 16 x 64; measured and not kept.)
  TILE_HALO_PS_192x64 = 23, TILE_HALO_PS_128x64 = 24, TILE_HALO_PS_128x128 = 25,
  // producer-specialised small generic tiles (the transformer's K = 320 .. 1280 projections: 160 blocks on 256 CUs, one block per
  // CU, no second block to overlap with): 4 + 4 and 4 + 8 wavefronts
  TILE_PS_64x64 = 26, TILE_PS2_64x64 = 27, TILE_PS_128x64 = 28, TILE_PS_64x128 = 29,
  // persistent GEGLU projection with EIGHT wavefronts per block (4 x 2, each 32 rows x 64 columns; round 5): a wavefront issues
  // in order -- its LDS-DMA requests (~130 cycles each), its MFMAs and its GELU arithmetic are one serial stream (1.8 k cycles per
  // K step for 512 of MFMA, 6.5 k of epilogue arithmetic per tile in the 4-wavefront form, whatever the co-resident block does:
  // the DF_PG_STAGGER experiment).  Twice the wavefronts halve every one of those streams.  31: 20 row-statistics slots (C = 1280)
  TILE_PGEGLU_128_W8 = 30, TILE_PGEGLU_128_W8L = 31,
  // WIDE GEGLU tiles (ffn_wide.hip, round 6): BM x 320 output columns (5 * 2^6: the model's channel counts are 5 * 2^k, so N = 8C
  // is always a multiple of 320 and (8192, 2560) / (2048, 5120) / (512, 10240) are EXACTLY 256 tiles of 256 / 128 / 64 rows -- one
  // per CU, no second round), 8 wavefronts as 4 (M) x 2 (N) on v_mfma_f32_16x16x32 fragments (80-column x / gate halves), two per
  // SIMD, one whole K panel per block, epilogue out of the accumulators.  Half the L2 -> LDS bytes per FLOP of the 128 x 128 tiles.
  TILE_WGEGLU_256 = 32, TILE_WGEGLU_128 = 33, TILE_WGEGLU_64 = 34, TILE_ALL = 35
};
static inline bool gemm_tile_is_pgeglu(int cfg) { return cfg == TILE_PGEGLU_128 || cfg == TILE_PGEGLU_64 || cfg == TILE_PGEGLU_128_W8 || cfg == TILE_PGEGLU_128_W8L; }
static inline bool gemm_tile_is_wgeglu(int cfg) { return cfg >= TILE_WGEGLU_256 && cfg <= TILE_WGEGLU_64; }
static inline bool gemm_tile_is_ps(int cfg) { return (cfg >= TILE_PS_256x128 && cfg <= TILE_PS2_128x128) || (cfg >= TILE_PS_64x64 && cfg <= TILE_PS_64x128); }
// ring depths (activation ring, weight ring) of the generic tiles; 0 for halo tiles
static inline void gemm_tile_rings(int cfg, int* nsta, int* nstb) {
  static const int a[TILE_ALL] = {4, 5, 5, 4, 4, 0, 0, 0, 3, 3, 2, 2, 2, 2, 2, 0, 0, 0, 3, 4, 4, 2, 3, 0, 0, 0, 4, 4, 4, 4, 2, 2, 2, 2, 3};
  static const int b[TILE_ALL] = {4, 5, 5, 4, 4, 0, 0, 0, 3, 3, 2, 2, 2, 2, 2, 0, 0, 0, 3, 4, 4, 2, 3, 0, 0, 0, 4, 4, 4, 4, 2, 2, 2, 2, 3};
  *nsta = a[cfg];
  *nstb = b[cfg];
}
// 64-column slots of row statistics a LayerNorm-folded GEMM can fold per row (gemm_impl.h LNS): C <= 1280
static inline int gemm_ln_max_slots() { return 20; }

static inline bool gemm_tile_is_halo(int cfg) {
  return (cfg >= TILE_HALO_128x64 && cfg <= TILE_HALO_128x128) || cfg == TILE_HALO_128x64_D || cfg == TILE_HALO_256x64_D ||
         cfg == TILE_HALO_192x64 || (cfg >= TILE_HALO_PS_192x64 && cfg <= TILE_HALO_PS_128x128);
}
// threads of a halo tile that issue its DMA requests (the LDS staging geometry follows from them: 8 threads per 128-B row)
static inline int gemm_halo_dma_threads(int cfg) {
  return (cfg == TILE_HALO_256x64 || cfg == TILE_HALO_256x64_D) ? 512 : 256;
}
static inline int gemm_halo_ring(int cfg) { return (cfg == TILE_HALO_128x64_D || cfg == TILE_HALO_256x64_D) ? 8 : 4; }

static inline void gemm_tile_dims(int cfg, int* bm, int* bn) {
  static const int d[TILE_ALL][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}, {32, 128},
                                     {128, 64},  {256, 64}, {128, 128}, {128, 256}, {256, 128},
                                     {128, 128}, {128, 64}, {64, 128}, {64, 64}, {32, 128},
                                     {128, 64}, {256, 64}, {192, 64}, {256, 128}, {128, 128}, {128, 128}, {128, 128}, {64, 128},
                                     {192, 64}, {128, 64}, {128, 128}, {64, 64}, {64, 64}, {128, 64}, {64, 128}, {128, 128}, {128, 128},
                                     {256, 320}, {128, 320}, {64, 320}};
  *bm = d[cfg][0];
  *bn = d[cfg][1];
}

// Can (tile, batch, splitk) run this problem?  (halo tiles: 3x3 stride-1 convs whose patch geometry fits LDS)
bool gemm_tile_valid(const GemmParams& p, int tile, int batch, int splitk);

// batch > 1 and splitk > 1 are mutually exclusive.
hipError_t launch_gemm(const GemmParams& p, int tile_cfg, int batch, hipStream_t stream);
// (32 x | 32 gate) GEGLU packing (rows of K operand values, column sums, folded bias) -> the wide tiles' 320-column packing (ffn_wide.hip)
hipError_t launch_pack_w320(const uint16_t* w, const float* cs, const float* bb, uint16_t* wo, float* cso, float* bbo, int N, int K, hipStream_t s);
