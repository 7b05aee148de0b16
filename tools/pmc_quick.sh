#!/bin/bash
# PMC HBM-side traffic of the UNet step only (two --pmc passes on a tuned configuration).  usage: tools/pmc_quick.sh [outdir]
ROOTD=$(pwd)
OUT=${1:-$ROOTD/gpurun_out/pmcq}
rm -rf $OUT; mkdir -p $OUT
export DF_TUNE_CACHE=$OUT/tune_cache.txt
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-modes --no-vae 2>/dev/null | tail -1 | cut -c1-120
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python $ROOTD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-modes --no-vae > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python $ROOTD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-modes --no-vae > /dev/null 2>&1
cd $ROOTD
python tools/pmc_traffic.py $OUT/pmc_fetch/f_counter_collection.csv $OUT/pmc_write/w_counter_collection.csv $OUT/pmc_traffic.json > /dev/null
rm -rf $OUT/pmc_fetch $OUT/pmc_write
python - <<PY
import json
t=json.load(open("$OUT/pmc_traffic.json"))
print({k:v for k,v in t.items() if k.endswith("per_step")})
PY
