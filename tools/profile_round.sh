#!/bin/bash
# Round profile on the GPU box (run from the repo root): bench line, per-op CSV, rocprofv3 kernel stats and the PMC
# HBM-traffic passes of the UNet step AND of the VAE decoder, all on the SAME tuned configuration (DF_TUNE_CACHE: the
# first run tunes and saves, the profiled runs load the choices so rocprof sees product launches only).
# Output: gpurun_out/profile/ ; copy what is to be judged into profiles/ (tools/profile_collect.py does it).
ROOTD=$(pwd)
OUT=$ROOTD/gpurun_out/profile
rm -rf $OUT; mkdir -p $OUT
# (round 6: bench.py runs the shipped plan table -- no tuning pass, rocprof sees product launches only; DF_TUNE_CACHE no longer needed)
python bench.py --dump-ops $OUT/ops_per_step.csv 2> $OUT/bench.err | tail -1 > $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $ROOTD/bench.py --steps 25 --no-cpu-baseline --no-modes --no-vae --no-batch8 > $OUT/rocprof_bench.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python $ROOTD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-modes --no-vae --no-batch8 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python $ROOTD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-modes --no-vae --no-batch8 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vae_stats -o s -- python $ROOTD/tools/vae_bench.py 10 > $OUT/vae_bench.txt 2>$OUT/vae_bench.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/vae_fetch -o f -- python $ROOTD/tools/vae_bench.py 4 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/vae_write -o w -- python $ROOTD/tools/vae_bench.py 4 > /dev/null 2>&1
cd $ROOTD
python tools/pmc_traffic.py $OUT/pmc_fetch/f_counter_collection.csv $OUT/pmc_write/w_counter_collection.csv $OUT/pmc_traffic.json > /dev/null
python tools/pmc_traffic.py $OUT/vae_fetch/f_counter_collection.csv $OUT/vae_write/w_counter_collection.csv $OUT/vae_pmc_traffic.json pack_latent_kernel > /dev/null
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/vae_stats -name "*kernel_stats.csv" -exec cp {} $OUT/vae_kernel_stats.csv \;
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/vae_fetch $OUT/vae_write $OUT/stats $OUT/vae_stats   # raw per-dispatch CSVs are large
python tools/e2e_bench.py > $OUT/e2e.txt 2>&1 || true
cut -c1-300 $OUT/bench.json; cat $OUT/vae_bench.txt; cat $OUT/pmc_traffic.json; cat $OUT/vae_pmc_traffic.json; tail -12 $OUT/e2e.txt
