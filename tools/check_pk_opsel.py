#!/usr/bin/env python
"""Build-time check for the packed-FP32 operand-select erratum found in round 6 (experiments/pk_opsel_probe.hip,
profiles/r6_pk_opsel_probe.txt): on MI355X a `v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32` whose LOW result takes the HIGH register
of src1 (`op_sel:[x,1(,x)]`) reads that operand as ZERO in lanes 48-63 while a wavefront of the same SIMD executes MFMAs.  The
compiler emits the form freely (`v[j].y += b.y` over two rows becomes `v_pk_add_f32 v[8:9], v[8:9], v[28:29] op_sel:[0,1]`); the
commuted form (`op_sel:[1,0]`: the select on src0) and a select on src2 are not affected.  Scans the gfx950 code objects of the
built libraries; exit code 1 when an affected encoding is present.

usage: tools/check_pk_opsel.py [-v] [lib ...]      (default: both in-tree libraries)"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
AFFECTED = ("v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32")


def code_objects(lib, tmp):
    dst = os.path.join(tmp, os.path.basename(lib))
    shutil.copy(lib, dst)
    subprocess.run([OBJDUMP, "--offloading", dst], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if f.startswith(os.path.basename(lib) + ".") and "gfx950" in f)


def affected(line):
    """True when `line` (one disassembled instruction) is a packed-FP32 op with the src1 operand select set."""
    op = line.split(None, 1)[0] if line.split() else ""
    if op not in AFFECTED:
        return False
    m = re.search(r"\bop_sel:\[([01](?:,[01])*)\]", line)
    if not m:
        return False
    bits = m.group(1).split(",")
    return len(bits) > 1 and bits[1] == "1"


def scan_text(text):
    """-> (number of packed-FP32 instructions, [(kernel, instruction)] with the affected encoding)"""
    kernel, n, bad = "?", 0, []
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            kernel = m.group(1)
            continue
        line = line.split("//")[0].strip()
        if line.startswith(AFFECTED):
            n += 1
            if affected(line):
                bad.append((kernel, line))
    return n, bad


def scan(co):
    out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    return scan_text(out)


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main(argv):
    verbose = "-v" in argv
    libs = [a for a in argv if a != "-v"] or [os.path.join(ROOT, "diff_foley_amd", n) for n in ("libdfengine.so", "libdfengine_f16.so")]
    if not os.path.exists(OBJDUMP):
        print("check_pk_opsel: llvm-objdump not found, nothing checked")
        return 0
    rc = 0
    for lib in libs:
        if not os.path.exists(lib):
            print(f"{lib}: missing")
            rc = 1
            continue
        with tempfile.TemporaryDirectory() as tmp:
            total, bad = 0, []
            for co in code_objects(lib, tmp):
                n, b = scan(co)
                total += n
                bad += b
        print(f"{os.path.basename(lib)}: {total} packed-FP32 instructions, {len(bad)} with the src1 operand select")
        if bad:
            rc = 1
            names = demangle(sorted({k for k, _ in bad}))
            per = {}
            for k, l in bad:
                per.setdefault(names[k], []).append(l)
            for k in sorted(per):
                print(f"  {len(per[k]):3d}  {k[:200]}")
                if verbose:
                    for l in per[k][:4]:
                        print("         " + l)
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
