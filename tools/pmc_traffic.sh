#!/bin/bash
# HBM-side traffic of the bench workload from PMC counters, collected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (with --kernel-trace only), FETCH_SIZE doubled (gfx950 reports half the
# bytes of wide coalesced reads; calibrated here on layernorm_kernel<5>: 8192x320 fp32 = 10.49 MB read -> 5.02 MB raw).
# Run on the GPU box from the repo root; writes gpurun_out/pmc_{fetch,write}/ and profiles/pmc_traffic.json.
set -e
ROOTD=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOTD/gpurun_out/pmc_fetch -o f -- python $ROOTD/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $ROOTD/gpurun_out/pmc_write -o w -- python $ROOTD/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $ROOTD
python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv gpurun_out/pmc_traffic.json
