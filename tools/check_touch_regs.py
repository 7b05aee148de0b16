#!/usr/bin/env python
"""Build-time check of the kernel-entry touch (csrc/common.h df_entry_touch) and of hand-issued loads in general.

Round 6 root cause of the "GEGLU epilogue race": the touch loads were issued from inline asm into a "+v" register; the compiler does
not know that a load is in flight to a register named only in inline asm, so it was free to copy the value away and re-use the
register while the load was still outstanding -- the late load then overwrote live data (request offsets, fragments, accumulators).
The touch is now made of ordinary (compiler-visible) global loads, for which the compiler's own s_waitcnt precedes any read or
overwrite of the destination.  This script keeps both properties checked in every built gfx950 code object:

  1. no `global_load_*` / `flat_load_*` / `buffer_load_*` (VGPR destination) appears in a kernel WITHOUT the compiler knowing it:
     we cannot see that from the binary directly, so the rule is on the SOURCE side -- csrc/ must not contain an inline-asm load with
     a VGPR destination (grep) -- and
  2. for every kernel that carries the touch (>= 2 single-dword global loads in front of the first operand request / barrier), every
     later instruction that WRITES a touch register or copies it must be preceded, since the last touch load, by an s_waitcnt vmcnt
     (any count: the compiler's scoreboard produced it for this register).  A violation means a hand-issued load crept back in.
  3. (performance, reported, not fatal) kernels in which that wait sits in front of the first LDS-DMA operand request: there the
     register allocator spilled a touch register in the prologue and the block now waits for a cold code line before its first request.

usage: tools/check_touch_regs.py [lib ...]      (default: both in-tree libraries); exit code 1 on a violation of 1 or 2."""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def dests(l):
    m = re.match(r"^(\S+)\s+(.*)$", l)
    if not m:
        return set()
    op, args = m.group(1), m.group(2)
    if op.startswith(("global_store", "buffer_store", "ds_write", "s_", "flat_store", "scratch_store", "v_cmp", "v_cmpx",
                      "v_accvgpr_write", "ds_append", "v_nop", "global_atomic", "buffer_atomic")):
        return set()
    if op.startswith("buffer_load") and l.rstrip().endswith("lds"):
        return set()
    first = args.split(",")[0].strip()
    mm = re.match(r"v\[(\d+):(\d+)\]$", first)
    if mm:
        return set(range(int(mm.group(1)), int(mm.group(2)) + 1))
    mm = re.match(r"v(\d+)$", first)
    return {int(mm.group(1))} if mm else set()


def source_rule():
    bad = []
    for path in sorted(glob.glob(os.path.join(ROOT, "diff_foley_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "diff_foley_amd", "csrc", "*.hip"))):
        for n, line in enumerate(open(path), 1):
            if "asm" in line and re.search(r"\"\s*(global|flat|buffer|scratch)_load_(dword|ubyte|ushort|short|sbyte)[^\"]*%0", line) and "lds" not in line:
                bad.append(f"{os.path.relpath(path, ROOT)}:{n}: inline-asm load with a register destination: {line.strip()[:120]}")
    return bad


def scan(lib):
    viol, early, ntouch = [], [], 0
    with tempfile.TemporaryDirectory() as tmp:
        dst = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, dst)
        subprocess.run([OBJDUMP, "--offloading", dst], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
        for co in sorted(f for f in glob.glob(dst + ".*") if "gfx950" in f):
            out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
            kern, kernels = None, {}
            for line in out.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
                if m:
                    kern = m.group(1)
                    kernels[kern] = []
                    continue
                line = line.split("//")[0].strip()
                if line and kern is not None:
                    kernels[kern].append(line)
            for k, ins in kernels.items():
                regs = {}
                for i, l in enumerate(ins[:200]):
                    if l.startswith("s_barrier") or (l.startswith("buffer_load") and l.endswith("lds")) or "vmcnt" in l:
                        break
                    m = re.match(r"global_load_dword v(\d+), v(\[\d+:\d+\]|\d+), (off|s\[\d+:\d+\])$", l)
                    if m:
                        regs.setdefault(int(m.group(1)), []).append(i)
                if sum(len(v) for v in regs.values()) < 2:
                    continue
                ntouch += 1
                first_dma = next((i for i, l in enumerate(ins) if l.startswith("buffer_load") and l.endswith("lds")), None)
                last_touch = max(max(v) for v in regs.values())
                for r, idx in regs.items():
                    waited = False
                    for i in range(idx[-1] + 1, len(ins)):
                        l = ins[i]
                        if "vmcnt" in l:
                            waited = True
                            if first_dma is not None and i < first_dma and i > last_touch:
                                early.append(f"{os.path.basename(lib)}: {k[:150]}: s_waitcnt vmcnt at instruction {i}, first operand request at {first_dma}")
                            break
                        if r in dests(l) or re.search(r"v_accvgpr_write_b32 a\d+, v%d$" % r, l) or re.search(r"v_mov_b32_e32 v\d+, v%d$" % r, l):
                            viol.append(f"{os.path.basename(lib)}: {k[:150]}: touch register v{r} (load at {idx[-1]}) is written / copied at {i} `{l}` with no s_waitcnt vmcnt in between")
                            break
    return ntouch, viol, sorted(set(early))


def main():
    libs = sys.argv[1:] or [os.path.join(ROOT, "diff_foley_amd", n) for n in ("libdfengine.so", "libdfengine_f16.so")]
    if not os.path.exists(OBJDUMP):
        print(f"check_touch_regs: {OBJDUMP} not found -- skipped (warning)")
        return 0
    bad = source_rule()
    tot = 0
    for lib in libs:
        n, viol, early = scan(lib)
        tot += n
        bad += viol
        for e in early:
            print("note (performance):", e)
    for b in bad:
        print("VIOLATION:", b)
    print(f"check_touch_regs: {tot} kernels with the entry touch scanned, {len(bad)} violation(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
