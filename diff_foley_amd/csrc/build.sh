#!/bin/bash
# Builds the engine (gfx950) in-tree, twice from the same sources:
#   diff_foley_amd/libdfengine.so      bf16 MFMA operands (default)
#   diff_foley_amd/libdfengine_f16.so  fp16 MFMA operands (-DDF_OPERAND_F16)
set -e
cd "$(dirname "$0")"
# -amdgpu-mfma-vgpr-form: MFMA accumulators live in architectural VGPRs (gfx950 has a unified register file), which
# removes the v_accvgpr_read/write copies around every accumulator touched by VALU code (attention rescale, epilogues):
# measured +5 % end to end, no spills in any kernel.
# -amdgpu-kernarg-preload-count=16: kernels whose leading arguments are scalars / pointers (GroupNorm, attention, the elementwise
# family -- not the GEMMs, whose one argument is a struct) get them in SGPRs at wavefront launch instead of through s_load from the
# cold scalar cache: GroupNorm family 0.596 -> 0.571 ms per step, +0.6 % end to end on the same box (round 4).
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-kernarg-preload-count=16"
SRCS="gemm gemm_m0a gemm_m0b gemm_m1 gemm_m2 gemm_m3 gemm_halo gemm_ps gemm_ps2 ffn ffn_wide elementwise attention backward cavp vocoder diag engine"
mkdir -p build/bf16 build/f16
pids=()
for v in bf16 f16; do
  DEF=""; [ $v = f16 ] && DEF="-DDF_OPERAND_F16"
  for f in $SRCS; do
    o=build/$v/$f.o
    if [ ! -f $o ] || [ $f.hip -nt $o ] || [ common.h -nt $o ] || [ gemm.h -nt $o ] || [ gemm_impl.h -nt $o ] || [ kernels.h -nt $o ] || [ ../../include/df_engine.h -nt $o ] || [ build.sh -nt $o ]; then
      hipcc $FLAGS $DEF -c $f.hip -o $o &
      pids+=($!)
    fi
  done
done
for p in "${pids[@]}"; do wait $p; done
for v in bf16 f16; do
  OUT=../libdfengine.so; [ $v = f16 ] && OUT=../libdfengine_f16.so
  objs=""; for f in $SRCS; do objs="$objs build/$v/$f.o"; done
  hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $OUT
  echo "built $(realpath $OUT)"
done
