#!/bin/bash
# Round-5 evidence run (GPU box, repo root): bench / rocprofv3 stats / PMC traffic of the UNet step and the VAE decoder, the SQ
# counter passes, the clock stamps of the small GEMM prologue and of the persistent GEGLU kernel (4- and 8-wavefront forms), the
# folded-skip conv shapes, the two-rank self-launch and its test.  Outputs under gpurun_out/; copy what is to be judged to profiles/.
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
bash tools/sq_counters.sh gpurun_out/sq > gpurun_out/sq.log 2>&1
{
  echo "# tools/gemm_stamps.py <tile> M N K (DF_GEMM_DBG=64): shader-clock stamps of the generic GEMM kernel, warm loop, one MI355X (round 5)."
  echo "# columns: entry -> operand requests of the prologue issued -> first tile landed -> K loop done | epilogue (cycles after the loop): barrier+park, pre-add, stores done"
  for a in "26 8192 320 320" "26 512 1280 1280" "26 2048 640 640"; do python tools/gemm_stamps.py $a 2>&1 | grep -v amdgpu; done
} > gpurun_out/r5_gemm_stamps.txt
{
  echo "# tools/pgeglu_stamps.py <tile>: persistent GEGLU kernel, st.ff1 at M = 8192: tile 21 (4 wavefronts) then tile 30 (8 wavefronts)"
  python tools/pgeglu_stamps.py 21 2>&1 | grep -v amdgpu
  python tools/pgeglu_stamps.py 30 2>&1 | grep -v amdgpu
  echo "# tools/pgeglu_probe.py 18,21,22,30,31 dbg: isolated times (us), /dN = debug switches (bit 0 no MFMA, 1 no stores, 2 no GELU, 3 no requests, 4 no fragment reads, 5 no statistics loads)"
  python tools/pgeglu_probe.py 18,21,22,30,31 dbg 2>&1 | grep -v amdgpu
} > gpurun_out/r5_pgeglu_stamps.txt
{
  echo "# tools/gemm_stamps_cold.py <tile> M N K: the same stamps, warm (third launch in a row) and COLD (behind a 2 GiB sweep of the caches,"
  echo "# another kernel's code and fresh operands -- the state a launch finds in the plan); three cold repetitions each"
  for a in "26 512 1280 1280" "28 512 3840 1280" "26 8192 320 320" "3 8192 320 320" "26 2048 640 640"; do echo "## tile $a"; python tools/gemm_stamps_cold.py $a 2>&1 | grep -v amdgpu; done
} > gpurun_out/r5_gemm_stamps_cold.txt
python tools/gemm_bench.py skip 2>&1 | grep -v amdgpu > gpurun_out/r5_skip_conv_bench.txt
DF_DIST_SHARE_GPU0=1 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5_bench_2rank_shared_gpu.json
python -m pytest tests/test_multi_rank_gpu.py tests/test_bench_selflaunch_gpu.py -q 2>&1 | tail -3 > gpurun_out/r5_two_rank_gpu_test.log
tail -3 gpurun_out/profile_round.log; tail -3 gpurun_out/sq.log; tail -3 gpurun_out/r5_two_rank_gpu_test.log
