#!/bin/bash
# SQ counters of one GEMM shape: tools/gemm_pmc.sh <name> <gemm_one.py args...>; output gpurun_out/pmc_<name>.txt
ROOTD=$(pwd); NAME=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $ROOTD/gpurun_out/pmc_$NAME/a -o a -- python $ROOTD/tools/gemm_one.py "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $ROOTD/gpurun_out/pmc_$NAME/b -o b -- python $ROOTD/tools/gemm_one.py "$@" > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $ROOTD/gpurun_out/pmc_$NAME/c -o c -- python $ROOTD/tools/gemm_one.py "$@" > /dev/null 2>&1
cd $ROOTD
python - $NAME <<'PY'
import csv,sys,collections,glob
name=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'gpurun_out/pmc_{name}/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'gemm' not in k and 'conv3x3' not in k: continue
        acc[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
out=open(f'gpurun_out/pmc_{name}.txt','w')
for k,c in acc.items():
    print(k,file=out); print(k)
    for n,v in sorted(c.items()):
        s=f'  {n:32s} {sum(v)/len(v):16.0f}  (n={len(v)})'
        print(s,file=out); print(s)
PY
