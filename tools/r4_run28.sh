#!/bin/bash
bash tools/ab2.sh "no_ps|DF_TILE_SKIP=0x1c0000" "ps|DF_X=1" "ps1_only|DF_TILE_SKIP=0x100000"
DF_TUNE_LOG=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-modes --no-vae --dump-ops gpurun_out/p28_ops.csv 2>gpurun_out/p28_tune.log | tail -1 | cut -c1-400
python - <<'PY'
import csv,collections
rows=list(csv.DictReader(open('gpurun_out/p28_ops.csv')))
c=collections.Counter(); t=collections.Counter()
for r in rows:
    if r['M']!='0': c[r['tile']]+=1; t[r['tile']]+=float(r['ms'])*1000
print({k:(c[k],round(t[k])) for k in sorted(c,key=int)})
PY
