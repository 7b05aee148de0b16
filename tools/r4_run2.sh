#!/bin/bash
mkdir -p gpurun_out/p2
python -m pytest tests/test_path_gpu.py -q -x -k "hoist or corrector or mel_mae or facade or inpaint" 2>&1 | tail -15 > gpurun_out/p2/t1.log
python -m pytest tests/test_path_fp16_gpu.py -q -x -k "saturation or mel_mae" 2>&1 | tail -15 > gpurun_out/p2/t2.log
python -m pytest tests/test_bench_selflaunch_gpu.py tests/test_multi_rank_gpu.py -q -x 2>&1 | tail -15 > gpurun_out/p2/t3.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/p2/bench.json 2> gpurun_out/p2/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-modes --no-vae --no-time-hoist > gpurun_out/p2/bench_nohoist.json 2>> gpurun_out/p2/bench.err
tail -3 gpurun_out/p2/t1.log gpurun_out/p2/t2.log gpurun_out/p2/t3.log
python - <<'PY'
import json
for f in ("bench","bench_nohoist"):
    d=json.loads(open(f"gpurun_out/p2/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("kernel_launches_per_step"))
PY
