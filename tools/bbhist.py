#!/usr/bin/env python
"""Per-basic-block instruction histogram of one kernel in a `hipcc -save-temps` gfx950 .s file: which blocks of the main loop
carry how many VALU / MFMA / LDS instructions (what the compiler made of the source, e.g. if-converted masks or accumulator copies).

usage: tools/bbhist.py <file.s> <mangled kernel name or unique substring> [min block size]"""
import collections
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    pat = sys.argv[2]
    names = [m.group(1) for m in re.finditer(r"^(\S+):\s*(?:;.*)?$", s, re.M) if pat in m.group(1) and not m.group(1).startswith(".")]
    if len(names) != 1:
        sys.exit(f"{len(names)} kernels match {pat!r}: {names[:8]}")
    name = names[0]
    i = s.index("\n" + name + ":")
    j = s.index(".Lfunc_end", i)
    lo = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    blocks, cur, lab = [], [], "entry"
    for l in s[i:j].splitlines():
        l = l.strip()
        if not l or l.startswith(";"):
            continue
        if re.match(r"^\.LBB\S+:", l):
            blocks.append((lab, cur))
            cur, lab = [], l.split()[0]
            continue
        if l.startswith(".") or l.endswith(":"):
            continue
        cur.append(l.split(";")[0].strip())
    blocks.append((lab, cur))
    for lab, b in blocks:
        if len(b) < lo:
            continue
        c = collections.Counter(x.split()[0] for x in b)
        nv = sum(v for k, v in c.items() if k.startswith("v_") and "mfma" not in k)
        print(f"{lab:12s} {len(b):4d} instr, VALU {nv:4d}: " + " ".join(f"{k}:{v}" for k, v in c.most_common(14)))
    for m in re.finditer(re.escape(name) + r"\.(num_vgpr|num_agpr), (\d+)", s):
        print(m.group(1), m.group(2))


if __name__ == "__main__":
    main()
