"""GPU: plain `python bench.py --gpus 2 ...` (no launcher, WORLD_SIZE unset) must start its own two ranks and print ONE JSON
line with n_gpus == 2 -- the form the driver's bench command takes when it asks for N > 1.  The test box has one MI355X, so
both ranks share cuda:0 through the DF_DIST_SHARE_GPU0 hook (gloo transport); on an N-GPU node the same command runs one rank
per GPU over RCCL."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus2_launches_its_own_ranks():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["DF_DIST_SHARE_GPU0"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--no-autotune", "--no-cpu-baseline"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 2 * d["per_gpu"]) < 1e-2 * d["value"]
    per_rank = d["config"]["weight_distribution"]["per_rank"]
    assert [p["rank"] for p in per_rank] == [0, 1]
    # every rank reports the world it saw on its backend and the device it ran on (bench.py run_mode)
    for p in per_rank:
        assert p["rccl_world_seen"] == 2 and p["pg_backend"] in ("nccl", "gloo") and p["cuda_device"] == 0 and p["pci_bus_id"]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_2rank_shared_gpu.json"), "w") as f:
        f.write(lines[0] + "\n")
    assert d["config"]["global_batch"] == 2 * d["config"]["batch_per_gpu"]
