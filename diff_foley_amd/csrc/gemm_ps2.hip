// Instantiations of the implicit-GEMM kernel (gemm_impl.h) with PRODUCER-SPECIALISED blocks, MODE 0 and 1: the small tiles.
#include "gemm_impl.h"

hipError_t launch_gemm_ps_small(int mode, int tile_cfg, int epi, const GemmParams& p, int zdim, hipStream_t stream) {
#define DF_T(T, BM, BN, NST, PS)                                              \
  case T:                                                                \
    if (mode == 1) switch (epi) {                                        \
      case EPI_LEAN: return launch_cfg<BM, BN, 2, 2, NST, 1, EPI_LEAN, NST, PS>(p, zdim, stream); \
      case EPI_SPLITK: return launch_cfg<BM, BN, 2, 2, NST, 1, EPI_SPLITK, NST, PS>(p, zdim, stream); \
      case EPI_ANY: return launch_cfg<BM, BN, 2, 2, NST, 1, EPI_ANY, NST, PS>(p, zdim, stream); \
      default: return hipErrorInvalidValue;                              \
    }                                                                    \
    if (mode != 0) return hipErrorInvalidValue;                          \
    switch (epi) {                                                       \
      case EPI_LEAN: return launch_cfg<BM, BN, 2, 2, NST, 0, EPI_LEAN, NST, PS>(p, zdim, stream); \
      case EPI_SPLITK: return launch_cfg<BM, BN, 2, 2, NST, 0, EPI_SPLITK, NST, PS>(p, zdim, stream); \
      case EPI_GEGLU: return launch_cfg<BM, BN, 2, 2, NST, 0, EPI_GEGLU, NST, PS>(p, zdim, stream); \
      case EPI_PROD: return launch_cfg<BM, BN, 2, 2, NST, 0, EPI_PROD, NST, PS>(p, zdim, stream); \
      case EPI_LNC: return launch_cfg<BM, BN, 2, 2, NST, 0, EPI_LNC, NST, PS>(p, zdim, stream); \
      case EPI_ANY: return launch_cfg<BM, BN, 2, 2, NST, 0, EPI_ANY, NST, PS>(p, zdim, stream); \
      case EPI_XS: return launch_cfg<BM, BN, 2, 2, NST, 0, EPI_XS, NST, PS>(p, zdim, stream); \
      default: return hipErrorInvalidValue;                              \
    }
  switch (tile_cfg) {
    DF_T(TILE_PS_64x64, 64, 64, 4, 1)
    DF_T(TILE_PS2_64x64, 64, 64, 4, 2)
    DF_T(TILE_PS_128x64, 128, 64, 4, 1)
    DF_T(TILE_PS_64x128, 64, 128, 4, 1)
    default: return hipErrorInvalidValue;
  }
#undef DF_T
}
