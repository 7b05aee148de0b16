#!/bin/bash
# Round profile on the GPU box (run from the repo root): bench line, per-op CSV, rocprofv3 kernel stats and the PMC
# HBM-traffic passes, all on the SAME tuned configuration (DF_TUNE_CACHE: the first run tunes and saves, the profiled
# runs load the choices so rocprof sees product launches only).  Output: gpurun_out/profile/.
set -e
ROOTD=$(pwd)
OUT=$ROOTD/gpurun_out/profile
mkdir -p $OUT
export DF_TUNE_CACHE=$OUT/tune_cache.txt
rm -f $DF_TUNE_CACHE
python bench.py --dump-ops $OUT/ops_per_step.csv 2> $OUT/bench.err | tail -1 > $OUT/bench.json
python bench.py --no-cpu-baseline --precision fp16 2>> $OUT/bench.err | tail -1 > $OUT/bench_fp16.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $ROOTD/bench.py --steps 25 --no-cpu-baseline > $OUT/rocprof_bench.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python $ROOTD/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python $ROOTD/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $ROOTD
python tools/pmc_traffic.py $OUT/pmc_fetch/f_counter_collection.csv $OUT/pmc_write/w_counter_collection.csv $OUT/pmc_traffic.json
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/pmc_fetch $OUT/pmc_write       # raw per-dispatch counter CSVs are large; the JSON summary is kept
find $OUT/stats -name "*kernel_trace.csv" -delete
python tools/e2e_bench.py > $OUT/e2e.txt 2>&1 || true
tail -3 $OUT/bench.json | cut -c1-400
