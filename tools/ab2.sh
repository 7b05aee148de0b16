#!/bin/bash
# Same-box A/B of bench.py (short form) across engine builds / env switches: tools/ab2.sh "label|ENV=.. ENV=.." ...   two interleaved passes
for pass in 1 2; do
  for arm in "$@"; do
    label=${arm%%|*}; envs=${arm#*|}
    env $envs python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-modes --no-vae --no-time-hoist 2>/dev/null | tail -1 > /tmp/_ab.json
    python - "$label" "$pass" <<'PY'
import json,sys
d=json.load(open('/tmp/_ab.json'))
k=d.get("kernel_ms_per_step",{})
print(f"{sys.argv[1]:28s} pass{sys.argv[2]}: {d['value']:7.2f} steps/s  {d['ms_per_step']:.3f} ms  gemm {k.get('gemm')} attn {k.get('attention')} gn {k.get('groupnorm')}")
PY
  done
done
