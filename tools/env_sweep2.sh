#!/bin/bash
# usage: tools/env_sweep2.sh "A=1 B=2" "A=2 B=3" ... ; one bench.py run per env assignment string
for v in "$@"; do
  env $v python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/_b.json
  python - "$v" <<'PY'
import json,sys
d=json.load(open('/tmp/_b.json'))
print(sys.argv[1], d["value"], d["ms_per_step"], d.get("kernel_ms_per_step",{}).get("gemm"))
PY
done
