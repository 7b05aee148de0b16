"""Per-kernel SQ counters of the denoise step from two rocprofv3 --pmc passes (tools/sq_counters.sh).

Only the dispatches of the last 4 full denoise steps are used (a step starts at pack_latent_bcast_kernel = t.lookup + x.pack), so tuning /
packing launches never enter.  Units (checked on this chip against kernel durations): SQ_BUSY_CYCLES is summed over the 32
shader engines, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (32 per SE; = MFMA instructions x 32 clk for 32x32x16),
SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT over the 256 CUs (8 per SE), SQ_WAVE_CYCLES / SQ_WAIT_* count 4-cycle quanta per wave.
  mfma_busy        = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES)      fraction of SIMD-cycles the matrix pipe is busy
  lds_busy         = SQ_LDS_IDX_ACTIVE / (8 * SQ_BUSY_CYCLES)
  lds_conflict     = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wait_any         = SQ_WAIT_ANY / SQ_WAVE_CYCLES                          wave parked at s_waitcnt / s_barrier
  wait_inst_lds    = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES
  waves_per_simd   = 4 * SQ_WAVE_CYCLES / (32 * SQ_BUSY_CYCLES)            mean resident waves per SIMD while the kernel runs
usage: sq_counters.py <dir with a/ and b/> out.json"""
import collections
import csv
import glob
import json
import re
import sys

MARK = "pack_latent_bcast_kernel"      # first kernel of a hoisted step since round 5 (table look-up + latent packing in one launch); was bcast_rows_kernel


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:90]


def load(d):
    files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    rows = [r for f in files for r in csv.DictReader(open(f))]
    if not rows:
        raise SystemExit(f"no counter rows under {d} (files: {files})")
    byd = collections.defaultdict(dict)
    names = {}
    for r in rows:
        did = int(r["Dispatch_Id"])
        byd[did][r["Counter_Name"]] = byd[did].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        names[did] = r["Kernel_Name"]
    ids = sorted(byd)
    marks = [i for i in ids if MARK in names[i]]
    steps = 4
    if len(marks) >= steps + 1:
        lo, hi = marks[-steps - 1], marks[-1]
        ids = [i for i in ids if lo <= i < hi]
    else:
        steps = None
    return [(names[i], byd[i]) for i in ids], steps


def main():
    d, out = sys.argv[1], sys.argv[2]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    steps = None
    import os
    subs = ("a", "b", "c") if os.path.isdir(f"{d}/c") else ("a", "b")
    for sub in subs:
        rows, st = load(f"{d}/{sub}")
        steps = st or steps
        for name, c in rows:
            k = short(name)
            for cn, v in c.items():
                agg[k][f"{sub}:{cn}"] += v
            if sub == "a":
                cnt[k] += 1
    res = []
    for k, c in agg.items():
        busy_a, busy_b = c.get("a:SQ_BUSY_CYCLES", 0.0), c.get("b:SQ_BUSY_CYCLES", 0.0)
        if busy_a <= 0:
            continue
        wc = max(c.get("a:SQ_WAVE_CYCLES", 0.0), 1.0)
        lds_act = c.get("b:SQ_LDS_IDX_ACTIVE", 0.0)
        res.append({
            "kernel": k, "launches_per_step": cnt[k] / (steps or 1),
            "busy_us_per_step": busy_a / 32 / 2400.0 / (steps or 1),       # SE-cycles / 32 SEs / 2.4 GHz
            "mfma_busy": c.get("a:SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (32 * busy_a),
            "lds_busy": lds_act / (8 * busy_b) if busy_b else None,
            "lds_conflict": c.get("b:SQ_LDS_BANK_CONFLICT", 0.0) / lds_act if lds_act else None,
            "wait_any": c.get("a:SQ_WAIT_ANY", 0.0) / wc,
            "wait_inst_any": c.get("a:SQ_WAIT_INST_ANY", 0.0) / wc,
            "wait_inst_lds": c.get("a:SQ_WAIT_INST_LDS", 0.0) / wc,
            "waves_per_simd": 4 * wc / (32 * busy_a),
            "waves_per_launch": c.get("a:SQ_WAVES", 0.0) / max(cnt[k], 1),
            # issue-slot view (pass c, round 4): instructions per wavefront by class
            **{key: c[f"c:{cn}"] / max(c.get("a:SQ_WAVES", 1.0), 1.0) for cn, key in (
                ("SQ_INSTS_MFMA", "mfma_insts_per_wave"), ("SQ_INSTS_VALU_MFMA_MOPS_F16", "mfma_mops_f16_per_wave"),
                ("SQ_INSTS_VMEM_RD", "vmem_rd_insts_per_wave"), ("SQ_INSTS_VMEM_WR", "vmem_wr_insts_per_wave"),
                ("SQ_INSTS_SMEM", "smem_insts_per_wave"), ("SQ_INSTS_LDS", "lds_insts_per_wave")) if f"c:{cn}" in c},
            **({"valu_insts_per_wave": c["b:SQ_INSTS_VALU"] / max(c.get("a:SQ_WAVES", 1.0), 1.0)} if "b:SQ_INSTS_VALU" in c else {}),
            **({"salu_insts_per_wave": c["b:SQ_INSTS_SALU"] / max(c.get("a:SQ_WAVES", 1.0), 1.0)} if "b:SQ_INSTS_SALU" in c else {}),
        })
    if not res:
        raise SystemExit("no kernel carries SQ_BUSY_CYCLES: counters seen = %s" % sorted({k for c in agg.values() for k in c})[:20])
    res.sort(key=lambda r: -r["busy_us_per_step"])
    tot = sum(r["busy_us_per_step"] for r in res)
    fam = collections.defaultdict(lambda: [0.0, 0.0])
    for r in res:
        f = "gemm" if ("gemm_bf16" in r["kernel"] or "halo" in r["kernel"] or "splitk" in r["kernel"] or "geglu_persistent" in r["kernel"] or "geglu_wide" in r["kernel"]) else (
            "attention" if "attention" in r["kernel"] else ("groupnorm" if "groupnorm" in r["kernel"] else "other"))
        fam[f][0] += r["busy_us_per_step"]
        fam[f][1] += r["mfma_busy"] * r["busy_us_per_step"]
    outd = {"what": __doc__.split("usage")[0].strip(), "steps_averaged": steps, "busy_us_per_step_all_kernels": tot,
            "families": {k: {"busy_us_per_step": v[0], "mfma_busy": v[1] / v[0] if v[0] else None} for k, v in fam.items()},
            "kernels": [{kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in r.items()} for r in res[:14]]}
    json.dump(outd, open(out, "w"), indent=1)
    for r in res[:8]:
        print(f"{r['busy_us_per_step']:8.1f} us/step  mfma {r['mfma_busy']:.3f}  lds {r['lds_busy'] or 0:.3f} confl {r['lds_conflict'] or 0:.3f}  "
              f"wait {r['wait_any']:.2f}  w/simd {r['waves_per_simd']:.2f}  {r['kernel'][:70]}")
    print({k: {a: round(b, 4) for a, b in v.items()} for k, v in outd["families"].items()})


main()
