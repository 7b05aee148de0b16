import collections
import csv
import json
import sys


def family(name):
    if "gemm_bf16" in name or "halo" in name or "splitk" in name:
        return "gemm"
    if "attention" in name:
        return "attention"
    if "groupnorm" in name:
        return "groupnorm"
    if "layernorm" in name:
        return "layernorm"
    return "other"


def agg(path, counter, last_launches):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    rows = rows[-last_launches:]            # steady state: the instrumented + timed steps at the end of the run
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in rows:
        f = family(r["Kernel_Name"])
        tot[f] += float(r["Counter_Value"]) * 1024.0       # counters are in KB
        n[f] += 1
    return tot, n


fetch, nf = agg(sys.argv[1], "FETCH_SIZE", 4000)
write, nw = agg(sys.argv[2], "WRITE_SIZE", 4000)
out = {"note": "bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (separate --pmc passes, last 4000 dispatches of bench.py)"}
for f in fetch:
    out[f + "_bytes_per_launch"] = round((2.0 * fetch[f] / max(1, nf[f])) + write[f] / max(1, nw[f]))
    out[f + "_launches"] = nf[f]
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
