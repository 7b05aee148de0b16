#!/bin/bash
# Round-6 evidence run (GPU box, repo root): bench / rocprofv3 stats / PMC traffic of the UNet step and the VAE decoder on the SHIPPED
# plan table (no tuning pass anywhere), the SQ counter passes, the wide GEGLU kernel's clock stamps and debug-switch timings beside the
# persistent kernel's, the reproducibility run of the cost-model plan (the round-6 root cause), the two-rank self-launch and its
# test.  Outputs under gpurun_out/; copy what is to be judged to profiles/.
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
bash tools/sq_counters.sh gpurun_out/sq > gpurun_out/sq.log 2>&1
{
  echo "# tools/pgeglu_stamps.py <tile> 0 M K (DBG build, p.dbg bit 6): shader-clock stamps per block -- entry -> operand requests of the prologue issued"
  echo "# -> first stage landed, then one column per K step, GELU epilogue, stores.  Tiles 32 / 33 / 34 = wide GEGLU (ffn_wide.hip), 30 = persistent 128 x 128"
  for a in "32 0 8192 320" "33 0 2048 640" "34 0 512 1280" "30 0 8192 320"; do echo "## tile $a"; python tools/pgeglu_stamps.py $a 2>&1 | grep -v amdgpu | grep "block\|entry" | head -8; done
  echo "# tools/pgeglu_probe.py 18,30,31,32,33,34 dbg, weights cold (cycled): isolated times (us), /dN = debug switches (bit 0 no MFMA, 1 no stores, 2 no GELU, 3 no requests, 4 no fragment reads, 5 no statistics loads)"
  (cd tools; python pgeglu_probe.py 18,30,31,32,33,34 dbg 2>&1 | grep -v amdgpu)
  echo "# the same, ONE weight buffer (warm)"
  (cd tools; PROBE_WARM=1 python pgeglu_probe.py 18,30,31,32,33,34 2>&1 | grep -v amdgpu)
} > gpurun_out/r6_wgeglu_stamps.txt
{
  echo "# tools/race_hunt.py 12 (full bf16 model, CFG forward x 12 under per-op workspace checksums): cost-model plans, then the shipped table"
  DF_TUNED_DEFAULTS=0 python tools/race_hunt.py 12 2>&1 | grep -v amdgpu | tail -3
  python tools/race_hunt.py 12 2>&1 | grep -v amdgpu | tail -3
  echo "# tools/nan_probe.py (cost-model plan, bf16): non-finite operand counts per op over three chained forwards"
  DF_TUNED_DEFAULTS=0 python tools/nan_probe.py 2>&1 | grep -v amdgpu | tail -3
} > gpurun_out/r6_reproducibility.txt
DF_DIST_SHARE_GPU0=1 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r6_bench_2rank_shared_gpu.json
python -m pytest tests/test_multi_rank_gpu.py tests/test_bench_selflaunch_gpu.py -q 2>&1 | tail -3 > gpurun_out/r6_two_rank_gpu_test.log
tail -3 gpurun_out/profile_round.log; tail -3 gpurun_out/sq.log; tail -3 gpurun_out/r6_two_rank_gpu_test.log; cat gpurun_out/r6_reproducibility.txt
