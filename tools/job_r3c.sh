set -x
mkdir -p gpurun_out/r3c
python -m pytest tests/test_multi_rank_gpu.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -6 > gpurun_out/r3c/two_rank.log
for v in v2 v0; do
  echo "--- $v conv" >> gpurun_out/r3c/gemm_bench.txt
  DF_LIB_OVERRIDE=$PWD/ab/libdf_$v.so python tools/gemm_bench.py conv 2>&1 | grep -v amdgpu >> gpurun_out/r3c/gemm_bench.txt
  echo "--- $v lin" >> gpurun_out/r3c/gemm_bench.txt
  DF_LIB_OVERRIDE=$PWD/ab/libdf_$v.so python tools/gemm_bench.py lin 2>&1 | grep -v amdgpu >> gpurun_out/r3c/gemm_bench.txt
done
bash tools/ab.sh "v2|DF_LIB_OVERRIDE=$PWD/ab/libdf_v2_f16.so" "v0|DF_LIB_OVERRIDE=$PWD/ab/libdf_v0_f16.so" "r2|DF_LIB_OVERRIDE=$PWD/ab/libdf_r2_f16.so" > gpurun_out/r3c/ab.txt 2>&1
# decomposition of every op of the step: full / no epilogue stores (2) / no main loop (4) / neither (6) / K tiles re-read tile 0 (1)
export DF_TUNE_CACHE=$PWD/gpurun_out/r3c/tc.txt
python bench.py --no-cpu-baseline --no-modes --no-vae --dump-ops gpurun_out/r3c/ops_dbg0.csv > gpurun_out/r3c/bench_dbg0.json 2>/dev/null
for d in 2 4 6 1; do
  DF_GEMM_DBG=$d python bench.py --no-cpu-baseline --no-modes --no-vae --dump-ops gpurun_out/r3c/ops_dbg$d.csv > gpurun_out/r3c/bench_dbg$d.json 2>/dev/null
done
