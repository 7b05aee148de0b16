"""Build-time guard against the packed-FP32 operand-select erratum (round 6; csrc/common.h pk_add_hi,
experiments/pk_opsel_probe.hip, profiles/r6_pk_opsel_probe.txt), CPU side: the encoding classifier of tools/check_pk_opsel.py and
the scan of the built libraries."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("check_pk_opsel", os.path.join(ROOT, "tools", "check_pk_opsel.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_encoding_classifier():
    t = _tool()
    # the forms the probe found broken: low result lane from the HIGH register of src1
    assert t.affected("v_pk_add_f32 v[8:9], v[8:9], v[28:29] op_sel:[0,1]")
    assert t.affected("v_pk_mul_f32 v[16:17], v[6:7], v[14:15] op_sel:[1,1] op_sel_hi:[1,0]")
    assert t.affected("v_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[2:3] op_sel:[0,1,0] op_sel_hi:[1,1,1]")
    assert t.affected("v_pk_add_f32 v[6:7], v[12:13], v[6:7] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]")
    # the forms it found sound: select on src0 / src2, high-lane selects, no select
    assert not t.affected("v_pk_add_f32 v[8:9], v[28:29], v[8:9] op_sel:[1,0] op_sel_hi:[1,1]")
    assert not t.affected("v_pk_fma_f32 v[8:9], v[28:29], v[40:41], v[8:9] op_sel:[1,0,0] neg_lo:[1,0,0] neg_hi:[1,0,0]")
    assert not t.affected("v_pk_fma_f32 v[2:3], v[2:3], v[4:5], v[6:7] op_sel:[0,0,1] op_sel_hi:[1,1,1]")
    assert not t.affected("v_pk_add_f32 v[14:15], v[14:15], v[28:29] op_sel_hi:[1,0]")
    assert not t.affected("v_pk_mul_f32 v[14:15], s[26:27], v[6:7]")
    assert not t.affected("v_pk_mov_b32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]")
    assert not t.affected("v_add_f32_e32 v8, v29, v8")


def test_scan_attributes_findings_to_kernels():
    t = _tool()
    text = """
0000000000001000 <kernel_a>:
	v_pk_add_f32 v[8:9], v[8:9], v[28:29] op_sel:[0,1]         // 000000001000: D3B24008 1802391C
	v_pk_mul_f32 v[2:3], v[4:5], v[6:7]
0000000000002000 <kernel_b>:
	v_pk_fma_f32 v[8:9], v[28:29], v[40:41], v[8:9] op_sel:[1,0,0]
"""
    n, bad = t.scan_text(text)
    assert n == 3 and bad == [("kernel_a", "v_pk_add_f32 v[8:9], v[8:9], v[28:29] op_sel:[0,1]")]


def test_built_libraries_hold_no_affected_encoding():
    t = _tool()
    libs = [os.path.join(ROOT, "diff_foley_amd", n) for n in ("libdfengine.so", "libdfengine_f16.so")]
    if not all(os.path.exists(p) for p in libs) or not os.path.exists(t.OBJDUMP):
        import pytest
        pytest.skip("libraries not built yet / llvm-objdump absent")
    assert t.main(libs) == 0
