"""Build-time guards of the kernel-entry touch (csrc/common.h df_entry_touch), CPU side.

Round 6 root cause of the timing-dependent garbage behind the cost-model plan's st.ffproj: touch loads issued from inline asm into a
register the compiler was free to vacate and re-use.  The touch is made of compiler-visible loads now; these tests keep it that way:
the source rule (no inline-asm load with a register destination anywhere in csrc/) and the code-object scan of the built libraries."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("check_touch_regs", os.path.join(ROOT, "tools", "check_touch_regs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_no_inline_asm_load_with_a_register_destination():
    assert _tool().source_rule() == []


def test_source_rule_recognises_the_round5_idiom(tmp_path, monkeypatch):
    t = _tool()
    bad = 'if (lane < 6) asm volatile("global_load_dword %0, %1, off" : "+v"(v) : "v"(ka) : "memory");\n'
    ok = 'asm volatile("global_store_dwordx4 %0, %1, off sc1\\n\\ts_nop 1" ::"v"(p), "v"(t) : "memory");\n'
    d = tmp_path / "diff_foley_amd" / "csrc"
    d.mkdir(parents=True)
    (d / "x.h").write_text(bad + ok)
    monkeypatch.setattr(t, "ROOT", str(tmp_path))
    found = t.source_rule()
    assert len(found) == 1 and "x.h:1" in found[0]


def test_destination_parser():
    t = _tool()
    assert t.dests("v_accvgpr_write_b32 a1, v5") == set()
    assert t.dests("ds_read_b128 v[74:77], v244") == {74, 75, 76, 77}
    assert t.dests("v_mov_b32_e32 v127, v66") == {127}
    assert t.dests("buffer_load_dwordx4 v7, s[60:63], 0 offen lds") == set()
    assert t.dests("global_store_dwordx2 v0, v[2:3], s[6:7]") == set()


def test_built_libraries_pass_the_scan():
    t = _tool()
    libs = [os.path.join(ROOT, "diff_foley_amd", n) for n in ("libdfengine.so", "libdfengine_f16.so")]
    if not all(os.path.exists(p) for p in libs) or not os.path.exists(t.OBJDUMP):
        import pytest
        pytest.skip("libraries not built yet / llvm-objdump absent")
    n, viol, _ = t.scan(libs[0])
    assert n > 100 and viol == []
