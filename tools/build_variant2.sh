#!/bin/bash
# Builds ONE variant of the engine for same-box A/Bs: tools/build_variant2.sh <name> <bf16|f16> [-DFLAG ...]
# -> diff_foley_amd/csrc/ab/lib_<name>_<type>.so (select with DF_LIB_OVERRIDE; ab/ travels to the GPU box, stays out of git).
# Only the translation units listed in VSRCS (default: all) are recompiled with the flags; the rest are taken from the product build
# (or from the object directory BASE, e.g. BASE=build/var_rcp_bf16).
set -e
name=$1; shift
typ=$1; shift
cd "$(dirname "$0")/../diff_foley_amd/csrc"
DEF=""; [ $typ = f16 ] && DEF="-DDF_OPERAND_F16"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-kernarg-preload-count=16 $DEF $*"
SRCS="gemm gemm_m0a gemm_m0b gemm_m1 gemm_m2 gemm_m3 gemm_halo gemm_ps gemm_ps2 ffn ffn_wide elementwise attention backward cavp vocoder diag engine"
VSRCS=${VSRCS:-$SRCS}
d=build/var_${name}_$typ; mkdir -p $d ab
pids=()
for f in $VSRCS; do hipcc $FLAGS -c $f.hip -o $d/$f.o & pids+=($!); done
for p in "${pids[@]}"; do wait $p; done
objs=""
for f in $SRCS; do
  if [[ " $VSRCS " == *" $f "* ]]; then objs="$objs $d/$f.o"; else objs="$objs ${BASE:-build/$([ $typ = f16 ] && echo f16 || echo bf16)}/$f.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ab/lib_${name}_$typ.so
echo "built ab/lib_${name}_$typ.so"
