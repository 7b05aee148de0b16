#!/bin/bash
mkdir -p gpurun_out/p4
python -m pytest tests/test_kernels_gpu.py -q -x -k "ln_folded or cross_attention" 2>&1 | tail -4 > gpurun_out/p4/t1.log
python -m pytest tests/test_path_gpu.py tests/test_path_fp16_gpu.py -q -x 2>&1 | tail -4 >> gpurun_out/p4/t1.log
cat gpurun_out/p4/t1.log
DF_TUNE_LOG=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-modes --no-vae --dump-ops gpurun_out/p4/ops.csv > gpurun_out/p4/bench.json 2> gpurun_out/p4/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/p4/bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("kernel_ms_per_step"))
PY
