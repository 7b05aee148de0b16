// Instantiations of the implicit-GEMM kernel (gemm_impl.h), MODE 3: nearest-x2 upsample + conv3x3 evaluated as four 2x2-tap
// convolutions on the INPUT-resolution map, one per output phase (Y & 1, X & 1), with per-phase weights that are the sums of the
// 3x3 taps falling on the same input pixel (openai_unetmodel.py:100-119 Upsample; 16 instead of 36 multiply-adds per input pixel).
#include "gemm_impl.h"

hipError_t launch_gemm_m3(int tile_cfg, int epi, const GemmParams& p, int zdim, hipStream_t stream) {
#define DF_T(T, BM, BN, WGM, WGN, NST)                                     \
  case T:                                                                \
    switch (epi) {                                                       \
      case EPI_LEAN: return launch_cfg<BM, BN, WGM, WGN, NST, 3, EPI_LEAN>(p, zdim, stream); \
      case EPI_SPLITK: return launch_cfg<BM, BN, WGM, WGN, NST, 3, EPI_SPLITK>(p, zdim, stream); \
      case EPI_ANY: return launch_cfg<BM, BN, WGM, WGN, NST, 3, EPI_ANY>(p, zdim, stream); \
      default: return hipErrorInvalidValue;                              \
    }
  switch (tile_cfg) {
    DF_T(TILE_64x64, 64, 64, 2, 2, 4)
    DF_T(TILE_128x256, 128, 256, 2, 4, 3)
    DF_T(TILE_256x128, 256, 128, 4, 2, 3)
    DF_T(TILE_128x128_S, 128, 128, 2, 2, 2)
    DF_T(TILE_128x64_S, 128, 64, 2, 2, 2)
    DF_T(TILE_64x128_S, 64, 128, 2, 2, 2)
    DF_T(TILE_64x64_S, 64, 64, 2, 2, 2)
    default: return hipErrorInvalidValue;
  }
#undef DF_T
}
