#!/bin/bash
# HBM-side bytes (FETCH_SIZE / WRITE_SIZE, separate passes) of ONE GEMM / conv shape per tile: tools/gemm_traffic.sh <gemm_one.py args...>
ROOTD=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gt_f /tmp/gt_w
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/gt_f -o f -- python $ROOTD/tools/gemm_one.py "$@" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/gt_w -o w -- python $ROOTD/tools/gemm_one.py "$@" > /dev/null 2>&1
python - "$@" <<'PY'
import csv,sys,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('/tmp/gt_f','/tmp/gt_w'):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name']
            if 'gemm' not in k and 'conv3x3' not in k and 'splitk' not in k: continue
            acc[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,c in acc.items():
    # same unit handling as tools/pmc_traffic.py: FETCH_SIZE / WRITE_SIZE are KB on this part, fetch doubled (64 B units -> 128 B lines)
    fe=c.get('FETCH_SIZE',[0]); wr=c.get('WRITE_SIZE',[0])
    print(' '.join(sys.argv[1:]), '|', k, f"read {2*sum(fe)/len(fe)/1024:8.2f} MB  write {sum(wr)/len(wr)/1024:8.2f} MB  (n={len(fe)})")
PY
