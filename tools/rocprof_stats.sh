#!/bin/bash
# rocprofv3 kernel-trace stats of bench.py for one arm.  usage: tools/rocprof_stats.sh <label> [ENV=..]...
ROOTD=$(pwd); label=$1; shift
OUT=$ROOTD/gpurun_out/prof_$label; mkdir -p $OUT
export DF_TUNE_CACHE=$OUT/tune.txt
env "$@" python bench.py --no-cpu-baseline > $OUT/bench.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $ROOTD/bench.py --steps 25 --no-cpu-baseline > $OUT/rocprof_bench.json 2>/dev/null
cd $ROOTD
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/stats -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/stats
python - $OUT <<'PY'
import csv,sys,collections
out=sys.argv[1]
rows=list(csv.DictReader(open(out+'/kernel_stats.csv')))
fam=collections.defaultdict(lambda:[0,0.0])
def f(n):
    if 'splitk_reduce' in n: return 'splitk_reduce'
    if 'gemm_bf16' in n or 'conv3x3_halo' in n: return 'gemm'
    if 'attention' in n: return 'attention'
    if 'groupnorm' in n: return 'groupnorm'
    if 'layernorm' in n: return 'layernorm'
    if 'pack_' in n or 'at::' in n or 'rocclr' in n.lower(): return 'setup/other-lib'
    return 'other'
for r in rows:
    k=f(r['Name']); fam[k][0]+=int(r['Calls']); fam[k][1]+=float(r['TotalDurationNs'])
steps=28+25+3  # warmup 3 + 25 timed + 25 instrumented (+3) in bench
for k,(c,t) in sorted(fam.items(), key=lambda kv:-kv[1][1]):
    print(f"{k:16s} calls={c:6d} total_ms={t/1e6:8.2f} avg_us={t/c/1e3:6.2f}")
# trimmed gaps from the trace: total span vs sum of durations for the bench kernels
tr=list(csv.DictReader(open(out+'/kernel_trace.csv')))
tr=[r for r in tr if 'pack_' not in r['Kernel_Name']]
tr.sort(key=lambda r:int(r['Start_Timestamp']))
n=len(tr); seg=tr[n//2:n//2+4000]
span=int(seg[-1]['End_Timestamp'])-int(seg[0]['Start_Timestamp'])
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seg)
print(f"trace window of {len(seg)} kernels: span {span/1e6:.3f} ms, sum of durations {busy/1e6:.3f} ms, idle {100*(1-busy/span):.1f} %")
PY
