#!/bin/bash
# SQ counters of one attention shape: tools/attn_pmc.sh N heads D Tq Tk
ROOTD=$(pwd)
python tools/attn_one.py "$@" 50
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ap_a /tmp/ap_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d /tmp/ap_a -o a -- python $ROOTD/tools/attn_one.py "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d /tmp/ap_b -o b -- python $ROOTD/tools/attn_one.py "$@" > /dev/null 2>&1
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('/tmp/ap_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'attention' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
for n,v in sorted(acc.items()): print(f"  {n:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
