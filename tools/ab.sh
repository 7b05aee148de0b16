#!/bin/bash
# Same-box A/B of bench.py across engine builds / env switches.  usage: tools/ab.sh "label|ENV=.. ENV=.." ...
# (DF_LIB_OVERRIDE=<path to an older libdfengine.so> selects another build.)  Two interleaved passes per arm.
for pass in 1 2; do
  for arm in "$@"; do
    label=${arm%%|*}; envs=${arm#*|}
    env $envs python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/_ab.json
    python - "$label" "$pass" <<'PY'
import json,sys
d=json.load(open('/tmp/_ab.json'))
k=d.get("kernel_ms_per_step",{}); n=d.get("north_star_families") or {}
print(f"{sys.argv[1]:28s} pass{sys.argv[2]}: {d['value']:7.2f} steps/s  {d['ms_per_step']:.3f} ms  gemm {k.get('gemm')} attn {k.get('attention')} gn {k.get('groupnorm')} ln {k.get('layernorm')} other {k.get('other')}  st_ms {n.get('spatial_transformer',{}).get('ms_per_step')}")
PY
  done
done
