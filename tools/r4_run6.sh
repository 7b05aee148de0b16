#!/bin/bash
mkdir -p gpurun_out/p6
python -m pytest tests/test_kernels_gpu.py -q -x -k "ln_folded" 2>&1 | tail -5 > gpurun_out/p6/t1.log
cat gpurun_out/p6/t1.log
DF_TUNE_LOG=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vae --dump-ops gpurun_out/p6/ops.csv > gpurun_out/p6/bench.json 2> gpurun_out/p6/bench.err
grep 'e1:' gpurun_out/p6/bench.err | grep '_1_1_0_1_1_e1' | cut -c1-300
python - <<'PY'
import json
d=json.loads(open("gpurun_out/p6/bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("kernel_ms_per_step"))
print({k:(v.get("steps_per_s"), (v.get("parity_vs_reference") or {}).get("mel_mae")) for k,v in d.get("modes",{}).items()})
PY
grep 'st.ff1' gpurun_out/p6/ops.csv
bash tools/ab2.sh "head|DF_X=1" "no_pgeglu|DF_TILE_CAP=21"
