#!/usr/bin/env python
"""Build-time check behind csrc/gemm.h gemm_kernarg_touch(): a GEMM kernel reads up to DF_CODE_TOUCH x 4 KB of its own code object
behind the program counter, bounded by the address of df_code_object_tail, a zero-initialised variable of the same code object.
That bound is only a bound if the variable lies BEHIND .text in the loaded image (in .bss, the last allocated section).  This
script checks exactly that for every gfx950 code object of the built libraries that defines the symbol.  Exit code 1 otherwise."""
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
                tails = [int(m.group(1), 16) for m in re.finditer(r"\d+:\s+([0-9a-f]{16})\s+\d+\s+OBJECT\s+\S+\s+\S+\s+\d+\s+\S*df_code_object_tail", out)]
                if not tails:
                    continue
                n_obj += 1
                text_end = sec[".text"][0] + sec[".text"][1]
                bss = sec.get(".bss")
                alloc_end = max(a + s for a, s in sec.values() if a)
                for t in tails:
                    ok = bss is not None and bss[0] <= t < bss[0] + bss[1] and t >= text_end and t + 64 <= alloc_end
                    if not ok:
                        bad += 1
                        print(f"{os.path.basename(lib)}:{f}: df_code_object_tail at {t:#x} is not behind .text (ends {text_end:#x}) inside .bss {bss}")
    print(f"checked {n_obj} code objects: {bad} violation(s)")
    return 1 if bad or not n_obj else 0


if __name__ == "__main__":
    sys.exit(main())
