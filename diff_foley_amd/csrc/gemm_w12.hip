// Deep-weight-ring instantiations of the implicit-GEMM kernel (gemm_impl.h, NSTB > NST), MODE 1 / 2 (conv3x3).
#include "gemm_impl.h"

hipError_t launch_gemm_w12(int mode, int tile_cfg, int epi, const GemmParams& p, int zdim, hipStream_t stream) {
#define DF_E(BM, BN, WGM, WGN, NST, NSTB, MODE)                            \
    switch (epi) {                                                       \
      case EPI_LEAN: return launch_cfg<BM, BN, WGM, WGN, NST, MODE, EPI_LEAN, NSTB>(p, zdim, stream); \
      case EPI_SPLITK: return launch_cfg<BM, BN, WGM, WGN, NST, MODE, EPI_SPLITK, NSTB>(p, zdim, stream); \
      case EPI_ANY: return launch_cfg<BM, BN, WGM, WGN, NST, MODE, EPI_ANY, NSTB>(p, zdim, stream); \
      default: return hipErrorInvalidValue;                              \
    }
#define DF_T(T, BM, BN, WGM, WGN, NST, NSTB)                               \
  case T:                                                                \
    if (mode == 1) { DF_E(BM, BN, WGM, WGN, NST, NSTB, 1) }              \
    else if (mode == 2) { DF_E(BM, BN, WGM, WGN, NST, NSTB, 2) }         \
    return hipErrorInvalidValue;
  switch (tile_cfg) {
    DF_T(TILE_256x64_W, 256, 64, 4, 1, 2, 10)
    DF_T(TILE_128x64_W, 128, 64, 2, 2, 2, 12)
    DF_T(TILE_128x128_W, 128, 128, 2, 2, 2, 6)
    default: return hipErrorInvalidValue;
  }
#undef DF_T
#undef DF_E
}
