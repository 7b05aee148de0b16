#!/bin/bash
# Builds libdfengine.so (gfx950) in-tree: diff-foley_amd/libdfengine.so
set -e
cd "$(dirname "$0")"
OUT=../libdfengine.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p build
pids=()
for f in gemm elementwise attention backward engine; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ gemm.h -nt build/$f.o ] || [ kernels.h -nt build/$f.o ] || [ ../../include/df_engine.h -nt build/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC build/gemm.o build/elementwise.o build/attention.o build/backward.o build/engine.o -o $OUT
echo "built $(realpath $OUT)"
