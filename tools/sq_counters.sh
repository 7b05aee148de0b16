#!/bin/bash
# SQ counters of the whole denoise step per kernel (MFMA busy, LDS stalls, bank conflicts, occupancy), two --pmc passes of the
# SAME bench command on a tuned configuration (counters with --kernel-trace only, as MI355X_MICROARCH.md prescribes).
# usage (GPU box, repo root): tools/sq_counters.sh [outdir]  ->  <outdir>/sq_counters.json (copy to profiles/rN_sq_counters.json)
ROOTD=$(pwd)
OUT=${1:-$ROOTD/gpurun_out/sq}
case $OUT in /*) ;; *) OUT=$ROOTD/$OUT ;; esac
rm -rf $OUT; mkdir -p $OUT
# (round 6: bench.py runs the shipped plan table -- no tuning pass, rocprof sees product launches only; DF_TUNE_CACHE no longer needed)
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-modes --no-vae --no-batch8"
python bench.py $ARGS 2>/dev/null | tail -1 | cut -c1-160
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \
  --kernel-trace --output-format csv -d $OUT/a -o a -- python $ROOTD/bench.py $ARGS > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU \
  --kernel-trace --output-format csv -d $OUT/b -o b -- python $ROOTD/bench.py $ARGS > $OUT/b.log 2>&1
# pass c (round 4): instruction mix -- MFMA / VMEM / LDS-DMA issue beside the waits (names probed with rocprofv3 -L on the box;
# a counter this rocprofv3 does not know is dropped from the list instead of failing the pass)
WANT="SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM"
rocprofv3 -L 2>/dev/null > $OUT/counters_available.txt
HAVE=""
n=0
for c in $WANT; do
  if grep -qw "$c" $OUT/counters_available.txt && [ $n -lt 8 ]; then HAVE="$HAVE $c"; n=$((n+1)); fi
done
echo "pass c counters:$HAVE"
if [ -n "$HAVE" ]; then
  rocprofv3 --pmc $HAVE --kernel-trace --output-format csv -d $OUT/c -o c -- python $ROOTD/bench.py $ARGS > $OUT/c.log 2>&1
fi
cd $ROOTD
find $OUT -name '*counter_collection.csv' | head; tail -n 3 $OUT/a.log; tail -n 3 $OUT/b.log
python tools/sq_counters.py $OUT $OUT/sq_counters.json && rm -rf $OUT/a $OUT/b $OUT/c
