"""HBM-side traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected separately with
--kernel-trace only, as MI355X_MICROARCH.md prescribes).  FETCH_SIZE is doubled (gfx950 reports half the bytes of wide
coalesced reads; calibrated here: groupnorm_reg_kernel<8> on 8192x320 fp32 reads 10.49 MB and reports 4.26 MB raw).
Only the dispatches of the LAST `steps` denoise steps are used (the first dispatch of every step is the time-embedding
kernel), so trial launches of the autotuner and weight packing never enter the averages.
usage: pmc_traffic.py fetch.csv write.csv out.json [marker-kernel-substring]"""
import collections
import csv
import json
import sys


def family(name):
    if "gemm_bf16" in name or "halo" in name or "splitk" in name or "geglu_persistent" in name or "geglu_wide" in name:
        return "gemm"
    if "attention" in name:
        return "attention"
    if "groupnorm" in name:
        return "groupnorm"
    if "layernorm" in name:
        return "layernorm"
    return "other"


def load(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def steady(rows, marker, steps):
    """Dispatches of the last `steps` steps: from the steps-th last marker kernel to the end."""
    idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    if len(idx) < steps + 1:
        return rows[-4000:], None
    # a step starts at its marker; drop the incomplete tail after the last full step
    return rows[idx[-steps - 1]:idx[-1]], steps


marker = sys.argv[4] if len(sys.argv) > 4 else "pack_latent_bcast_kernel"      # t.lookup + x.pack: first kernel of a hoisted step (round 5)
out = {"note": "bytes per launch = 2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes, dispatches of the last full steps only",
       "marker_kernel": marker}
per_kernel = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
fam = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
for which, path, counter in ((0, sys.argv[1], "FETCH_SIZE"), (1, sys.argv[2], "WRITE_SIZE")):
    rows, steps = steady(load(path, counter), marker, 4)
    out["steps_used"] = steps
    for r in rows:
        b = float(r["Counter_Value"]) * 1024.0 * (2.0 if which == 0 else 1.0)
        for d, k in ((per_kernel, r["Kernel_Name"][:70]), (fam, family(r["Kernel_Name"]))):
            d[k][2 * which] += 1
            d[k][2 * which + 1] += b
for f, (nf, bf, nw, bw) in fam.items():
    out[f + "_bytes_per_launch"] = round(bf / max(1, nf) + bw / max(1, nw))
    out[f + "_launches_per_step"] = round(nf / (out["steps_used"] or 1), 1)
    out[f + "_bytes_per_step"] = round((bf / max(1, nf) + bw / max(1, nw)) * nf / (out["steps_used"] or 1))
out["kernels"] = {k: {"launches": nf, "read_MB": round(bf / max(1, nf) / 1e6, 2), "write_MB": round(bw / max(1, nw) / 1e6, 2)}
                  for k, (nf, bf, nw, bw) in sorted(per_kernel.items(), key=lambda kv: -(kv[1][1] + kv[1][3]))[:25]}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}))
