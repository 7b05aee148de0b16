"""GPU: plan-level rewrites that must not change a single bit.  The engine reads its A/B switches once per process, so every arm
runs in its own interpreter (tools/chk_probe.py --mode hash: tiny model, cond -> 6-step CFG DDIM -> decode, MD5 of the mel).

  DF_NO_GNOWN=1   the GroupNorm that follows a split-K GEMM does NOT take over that GEMM's reduce (csrc/elementwise.hip
                  GnSlabs::own): the hand-over sums the slabs in slab order and adds bias / residual in the reduce kernel's order,
                  so both arms must produce the same bytes;
  (the same run also repeats the sampling 6 times in-process: every repetition must give the same hash -- run-to-run determinism)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hashes(env_extra):
    env = dict(os.environ)
    # cost-model plans in both arms: the shipped plan table (round 6) is keyed by GEMM shape AND epilogue class, and the hand-over
    # changes the class of the GEMMs in front of a norm -- with the table on, the two arms may run different tiles / split-K factors
    # (another fp32 summation order), which is not what this test is about
    env["DF_TUNED_DEFAULTS"] = "0"
    env.update(env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "chk_probe.py"), "6", "--mode", "hash"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)["mel_hashes"]


def test_groupnorm_reduce_handover_changes_no_bit():
    a = _hashes({})
    b = _hashes({"DF_NO_GNOWN": "1"})
    assert len(a) == 1 and len(b) == 1, (a, b)          # 6 repetitions each: one hash per arm
    assert list(a) == list(b), (a, b)
