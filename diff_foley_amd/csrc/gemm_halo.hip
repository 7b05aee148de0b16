// Instantiations of the halo conv3x3 kernel (gemm_impl.h).
#include "gemm_impl.h"

hipError_t launch_gemm_halo(int tile_cfg, int epi, const GemmParams& p, int zdim, hipStream_t stream) {
#define DF_H(T, BM, BN, WGM, WGN, NSTW) DF_HP(T, BM, BN, WGM, WGN, NSTW, 0)
#define DF_HP(T, BM, BN, WGM, WGN, NSTW, PS)                                         \
  case T:                                                                            \
    switch (epi) {                                                                   \
      case EPI_LEAN: return launch_halo<BM, BN, WGM, WGN, NSTW, EPI_LEAN, PS>(p, zdim, stream);     \
      case EPI_SPLITK: return launch_halo<BM, BN, WGM, WGN, NSTW, EPI_SPLITK, PS>(p, zdim, stream); \
      case EPI_ANY: return launch_halo<BM, BN, WGM, WGN, NSTW, EPI_ANY, PS>(p, zdim, stream);       \
      default: return hipErrorInvalidValue;                                          \
    }
  switch (tile_cfg) {
    DF_H(TILE_HALO_128x64, 128, 64, 2, 2, 4)
    DF_H(TILE_HALO_256x64, 256, 64, 4, 2, 4)
    DF_H(TILE_HALO_128x128, 128, 128, 2, 2, 4)
    DF_H(TILE_HALO_128x64_D, 128, 64, 2, 2, 8)
    DF_H(TILE_HALO_256x64_D, 256, 64, 4, 2, 8)
    DF_H(TILE_HALO_192x64, 192, 64, 2, 2, 4)
    DF_HP(TILE_HALO_PS_192x64, 192, 64, 2, 2, 4, 1)
    DF_HP(TILE_HALO_PS_128x64, 128, 64, 2, 2, 4, 1)
    DF_HP(TILE_HALO_PS_128x128, 128, 128, 2, 2, 4, 1)
    default: return hipErrorInvalidValue;
  }
#undef DF_H
#undef DF_HP
}
