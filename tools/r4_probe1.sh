#!/bin/bash
# round-4 probe 1: baseline bench of the round-3 binary on this box + decomposition of the linear GEMM shapes
mkdir -p gpurun_out/p1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/p1/bench.json 2> gpurun_out/p1/bench.err
tail -c 600 gpurun_out/p1/bench.json
for d in 0 2 4 6; do
  echo "== DF_GEMM_DBG=$d"
  DF_GEMM_DBG=$d python tools/gemm_bench.py lin all 2>&1 | tee gpurun_out/p1/gemm_lin_dbg$d.txt | cut -c1-400
done
