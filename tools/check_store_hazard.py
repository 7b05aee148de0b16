#!/usr/bin/env python
"""Build-time check of the write-through store idiom (csrc/common.h st_wt): gfx950 needs 2 wait states between a VMEM store of
more than 64 bits and a VALU write of the store's data VGPRs, and the compiler's hazard recogniser cannot see a store that sits in
inline asm.  Scans the gfx950 code objects of the built libraries for every `global_store_dwordx4 ... sc1`: it must be followed by
`s_nop 1` (or more), or by two instructions that are not VALU writes of its data registers.  Exit code 1 on a violation.

usage: tools/check_store_hazard.py [lib ...]      (default: both in-tree libraries)"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def code_objects(lib, tmp):
    dst = os.path.join(tmp, os.path.basename(lib))
    shutil.copy(lib, dst)
    subprocess.run([OBJDUMP, "--offloading", dst], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if f.startswith(os.path.basename(lib) + ".") and "gfx950" in f)


def vregs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(co):
    out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    ins = []
    for line in out.splitlines():
        line = line.split("//")[0].strip()
        if not line or line.endswith(":") or line.startswith(("Disassembly", "/")):
            continue
        ins.append(line)
    n_store = bad = 0
    for i, l in enumerate(ins):
        if not (l.startswith("global_store_dwordx4") and "sc1" in l):
            continue
        n_store += 1
        ops = [t.strip() for t in l.split(None, 1)[1].split(",")]
        data = vregs(ops[1]) if len(ops) > 1 else set()
        states = 0
        for nxt in ins[i + 1:i + 3]:
            m = re.match(r"s_nop (\d+)", nxt)
            if m:
                states += int(m.group(1)) + 1
                if states >= 2:
                    break
                continue
            if nxt.startswith("v_") and not nxt.startswith("v_cmp"):
                dst = nxt.split(None, 1)[1].split(",")[0].strip()
                if vregs(dst) & data and states < 2:
                    bad += 1
                    print(f"HAZARD in {os.path.basename(co)}: `{l}` -> `{nxt}` after {states} wait state(s)")
                    break
            states += 1
            if states >= 2:
                break
    return n_store, bad


def main():
    libs = sys.argv[1:] or [os.path.join(ROOT, "diff_foley_amd", n) for n in ("libdfengine.so", "libdfengine_f16.so")]
    total = viol = 0
    for lib in libs:
        with tempfile.TemporaryDirectory() as tmp:
            for co in code_objects(lib, tmp):
                s, b = scan(co)
                total += s
                viol += b
    print(f"checked {total} dwordx4 sc1 stores in {len(libs)} libraries: {viol} hazard(s)")
    return 1 if viol else 0


if __name__ == "__main__":
    sys.exit(main())
