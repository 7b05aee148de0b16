// Instantiations of the implicit-GEMM kernel (gemm_impl.h), MODE 0: one translation unit per mode so they build in parallel.
#include "gemm_impl.h"

hipError_t launch_gemm_m0b(int tile_cfg, int epi, const GemmParams& p, int zdim, hipStream_t stream) {
#define DF_T(T, BM, BN, WGM, WGN, NST)                                     \
  case T:                                                                \
    switch (epi) {                                                       \
      case EPI_LEAN: return launch_cfg<BM, BN, WGM, WGN, NST, 0, EPI_LEAN>(p, zdim, stream); \
      case EPI_SPLITK: return launch_cfg<BM, BN, WGM, WGN, NST, 0, EPI_SPLITK>(p, zdim, stream); \
      case EPI_GEGLU: return launch_cfg<BM, BN, WGM, WGN, NST, 0, EPI_GEGLU>(p, zdim, stream); \
      case EPI_PROD: return launch_cfg<BM, BN, WGM, WGN, NST, 0, EPI_PROD>(p, zdim, stream); \
      case EPI_LNC: return launch_cfg<BM, BN, WGM, WGN, NST, 0, EPI_LNC>(p, zdim, stream); \
      case EPI_ANY: return launch_cfg<BM, BN, WGM, WGN, NST, 0, EPI_ANY>(p, zdim, stream); \
      case EPI_XS: return launch_cfg<BM, BN, WGM, WGN, NST, 0, EPI_XS>(p, zdim, stream); \
      default: return hipErrorInvalidValue;                              \
    }
  switch (tile_cfg) {
    DF_T(TILE_128x128_S, 128, 128, 2, 2, 2)
    DF_T(TILE_128x64_S, 128, 64, 2, 2, 2)
    DF_T(TILE_64x128_S, 64, 128, 2, 2, 2)
    DF_T(TILE_64x64_S, 64, 64, 2, 2, 2)
    DF_T(TILE_32x128_S, 32, 128, 1, 4, 2)
    default: return hipErrorInvalidValue;
  }
#undef DF_T
}
