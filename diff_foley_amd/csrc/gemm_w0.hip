// Deep-weight-ring instantiations of the implicit-GEMM kernel (gemm_impl.h, NSTB > NST), MODE 0 (Linear / 1x1).
#include "gemm_impl.h"

hipError_t launch_gemm_w0(int tile_cfg, int epi, const GemmParams& p, int zdim, hipStream_t stream) {
#define DF_T(T, BM, BN, WGM, WGN, NST, NSTB)                               \
  case T:                                                                \
    switch (epi) {                                                       \
      case EPI_LEAN: return launch_cfg<BM, BN, WGM, WGN, NST, 0, EPI_LEAN, NSTB>(p, zdim, stream); \
      case EPI_SPLITK: return launch_cfg<BM, BN, WGM, WGN, NST, 0, EPI_SPLITK, NSTB>(p, zdim, stream); \
      case EPI_GEGLU: return launch_cfg<BM, BN, WGM, WGN, NST, 0, EPI_GEGLU, NSTB>(p, zdim, stream); \
      case EPI_PROD: return launch_cfg<BM, BN, WGM, WGN, NST, 0, EPI_PROD, NSTB>(p, zdim, stream); \
      case EPI_LNC: return launch_cfg<BM, BN, WGM, WGN, NST, 0, EPI_LNC, NSTB>(p, zdim, stream); \
      case EPI_ANY: return launch_cfg<BM, BN, WGM, WGN, NST, 0, EPI_ANY, NSTB>(p, zdim, stream); \
      case EPI_XS: return launch_cfg<BM, BN, WGM, WGN, NST, 0, EPI_XS, NSTB>(p, zdim, stream); \
      default: return hipErrorInvalidValue;                              \
    }
  switch (tile_cfg) {
    DF_T(TILE_256x64_W, 256, 64, 4, 1, 2, 10)
    DF_T(TILE_128x64_W, 128, 64, 2, 2, 2, 12)
    DF_T(TILE_128x128_W, 128, 128, 2, 2, 2, 6)
    default: return hipErrorInvalidValue;
  }
#undef DF_T
}
