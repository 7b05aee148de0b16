#!/bin/bash
# Builds ONE fp16-operand variant of the engine for same-box A/Bs: tools/build_variant.sh <name> [-DFLAG ...]
# -> diff_foley_amd/csrc/ab/lib_<name>_f16.so (select with DF_LIB_OVERRIDE; ab/ travels to the GPU box, stays out of git).
set -e
name=$1; shift
cd "$(dirname "$0")/../diff_foley_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-kernarg-preload-count=16 -DDF_OPERAND_F16 $*"
SRCS="gemm gemm_m0a gemm_m0b gemm_m1 gemm_m2 gemm_m3 gemm_halo gemm_ps gemm_ps2 ffn elementwise attention backward cavp vocoder diag engine"
d=build/var_$name; mkdir -p $d ab
pids=()
for f in $SRCS; do hipcc $FLAGS -c $f.hip -o $d/$f.o & pids+=($!); done
for p in "${pids[@]}"; do wait $p; done
objs=""; for f in $SRCS; do objs="$objs $d/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ab/lib_${name}_f16.so
echo "built ab/lib_${name}_f16.so"
