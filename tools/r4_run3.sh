#!/bin/bash
mkdir -p gpurun_out/p3
python -m pytest tests/test_kernels_gpu.py -q -x -k "test_gemm or conv3x3 or ln_folded" 2>&1 | tail -8 > gpurun_out/p3/t1.log
python -m pytest tests/test_path_gpu.py -q -x -k "corrector" 2>&1 | tail -8 >> gpurun_out/p3/t1.log
cat gpurun_out/p3/t1.log
python tools/cold_probe.py 2>/dev/null | tee gpurun_out/p3/cold.txt
DF_TUNE_LOG=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-modes --no-vae --dump-ops gpurun_out/p3/ops.csv > gpurun_out/p3/bench.json 2> gpurun_out/p3/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/p3/bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("kernel_ms_per_step"))
PY
