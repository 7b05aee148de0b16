#!/bin/bash
# usage: tools/env_sweep.sh VAR v1 v2 ... ; runs bench.py per value and prints steps/s, ms/step, gemm ms
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/_b.json
  python - "$v" <<'PY'
import json,sys
d=json.load(open('/tmp/_b.json'))
print(sys.argv[1], d["value"], d["ms_per_step"], d.get("kernel_ms_per_step",{}).get("gemm"))
PY
done
