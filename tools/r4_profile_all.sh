#!/bin/bash
# Round-4 evidence run (GPU box, repo root): bench / rocprofv3 stats / PMC traffic of the UNet step and the VAE decoder, the SQ
# counter passes, the in-kernel clock stamps of the halo conv (symmetric tile 17 vs producer-specialised tile 23) and of the
# persistent GEGLU kernel, the two-rank self-launch and its test.  Outputs under gpurun_out/; tools/profile_collect.py copies.
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
bash tools/sq_counters.sh gpurun_out/sq > gpurun_out/sq.log 2>&1
{
  echo "# tools/halo_stamps.py <tile> 8 16 64 320 320 (res.conv1 of the 16 x 64 level): shader-clock stamps of two blocks"
  echo "## tile 17 (192 x 64, symmetric: every wavefront requests and multiplies)"
  python tools/halo_stamps.py 17 8 16 64 320 320 2>&1 | grep -v amdgpu | head -3
  echo "## tile 23 (192 x 64, producer-specialised: 4 consumer + 4 producer wavefronts)"
  python tools/halo_stamps.py 23 8 16 64 320 320 2>&1 | grep -v amdgpu | head -3
  echo "## tile 23, conv 640 -> 640 at 8 x 32"
  python tools/halo_stamps.py 23 8 8 32 640 640 2>&1 | grep -v amdgpu | head -2
} > gpurun_out/r4_halo_stamps.txt
python tools/pgeglu_stamps.py > gpurun_out/r4_pgeglu_stamps.txt 2>&1
python tools/gemm_bench.py "" all 2>&1 | grep -v amdgpu > gpurun_out/r4_gemm_bench_all.txt
DF_DIST_SHARE_GPU0=1 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4_bench_2rank_shared_gpu.json
python -m pytest tests/test_multi_rank_gpu.py tests/test_bench_selflaunch_gpu.py -q 2>&1 | tail -3 > gpurun_out/r4_two_rank_gpu_test.log
tail -3 gpurun_out/profile_round.log; tail -3 gpurun_out/sq.log; cat gpurun_out/r4_halo_stamps.txt | cut -c1-200; tail -3 gpurun_out/r4_two_rank_gpu_test.log
