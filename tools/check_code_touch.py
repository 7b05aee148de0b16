#!/usr/bin/env python
"""Build-time check behind csrc/common.h df_entry_touch(): a kernel reads up to DF_CODE_TOUCH x 4 KB of its own code object behind
the program counter, bounded by `_etext`, the linker's symbol for the end of the code object's .text section.  This script checks, for
every gfx950 code object of the built libraries that references the symbol, that it is defined and equals .text's end (so no touched
line can lie outside the mapped, executable section).  Exit code 1 otherwise."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"


def main():
    libs = sys.argv[1:] or [os.path.join(ROOT, "diff_foley_amd", n) for n in ("libdfengine.so", "libdfengine_f16.so")]
    n_obj = bad = 0
    for lib in libs:
        with tempfile.TemporaryDirectory() as tmp:
            dst = os.path.join(tmp, os.path.basename(lib))
            shutil.copy(lib, dst)
            subprocess.run([LLVM + "llvm-objdump", "--offloading", dst], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
            for f in sorted(os.listdir(tmp)):
                if "gfx950" not in f:
                    continue
                out = subprocess.run([LLVM + "llvm-readelf", "-S", "-s", "-W", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
                sec = {m.group(1): (int(m.group(2), 16), int(m.group(3), 16)) for m in
                       re.finditer(r"\]\s+(\.\S+)\s+\S+\s+([0-9a-f]{16})\s+[0-9a-f]+\s+([0-9a-f]+)", out)}
                ends = [int(m.group(1), 16) for m in re.finditer(r"\d+:\s+([0-9a-f]{16})\s+\d+\s+\S+\s+\S+\s+\S+\s+\d+\s+_etext\s*$", out, re.M)]
                if not ends:
                    continue
                n_obj += 1
                text_end = sec[".text"][0] + sec[".text"][1]
                for t in set(ends):
                    if t != text_end:
                        bad += 1
                        print(f"{os.path.basename(lib)}:{f}: _etext at {t:#x} is not the end of .text ({text_end:#x})")
    print(f"checked {n_obj} code objects: {bad} violation(s)")
    return 1 if bad or not n_obj else 0


if __name__ == "__main__":
    sys.exit(main())
