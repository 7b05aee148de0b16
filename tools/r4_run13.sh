#!/bin/bash
mkdir -p gpurun_out/p13
python -m pytest tests/test_kernels_gpu.py -q -x -k "fused_cross_attention or full_cross" 2>&1 | tail -4 > gpurun_out/p13/t1.log
python -m pytest tests/test_path_gpu.py tests/test_path_fp16_gpu.py -q -x 2>&1 | tail -4 >> gpurun_out/p13/t1.log
cat gpurun_out/p13/t1.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vae --dump-ops gpurun_out/p13/ops.csv > gpurun_out/p13/bench.json 2> gpurun_out/p13/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/p13/bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("kernel_ms_per_step"))
print({k:(v.get("steps_per_s"), (v.get("parity_vs_reference") or {}).get("mel_mae")) for k,v in d.get("modes",{}).items()})
PY
grep 'st.xatt\|st.xs\|st.xo' gpurun_out/p13/ops.csv
bash tools/ab2.sh "head|DF_X=1" "no_xfuse|DF_NO_XFUSE=1" "xfuse640|DF_XFUSE_MAXC=640"
