"""GPU: the collectives of diff_foley_amd/parallel.py and bench.py on their REAL backend.  The test box has one GPU, so this is a
world-size-1 RCCL group ("nccl" IS RCCL on ROCm): communicator creation on this image, and every collective call the N > 1 path
makes -- int64 size broadcast, uint8 payload broadcast (256 MB), all_gather of sizes, gather with a destination list of padded
shards, all_gather_object, barrier -- with the dtypes / devices it makes them with.  The multi-rank LOGIC (sharding, padded
gather, sidecar import) is covered on gloo (tests/test_parallel_cpu.py, tests/test_multi_rank_gpu.py); what is left for the first
multi-GPU run is xGMI itself."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from diff_foley_amd import parallel
r, w, local = parallel.init_process_group()            # world 1: no group yet (the helpers short-circuit)
assert (r, w, local) == (0, 1, 0)
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
assert dist.get_backend() == "nccl"
n = torch.tensor([123456789012], dtype=torch.int64, device=dev)
dist.broadcast(n, src=0)
assert int(n.item()) == 123456789012
blob = torch.arange(256 << 20, dtype=torch.int64, device=dev).to(torch.uint8)        # 256 MB payload, like the packed operand blob
ref = blob.clone()
dist.broadcast(blob, src=0)
assert torch.equal(blob, ref)
sizes = [torch.zeros(1, dtype=torch.int64, device=dev)]
dist.all_gather(sizes, torch.tensor([3], dtype=torch.int64, device=dev))
assert int(sizes[0].item()) == 3
mel = torch.randn(3, 128, 512, device=dev)
bufs = [torch.empty_like(mel)]
dist.gather(mel.contiguous(), bufs, dst=0)
assert torch.equal(bufs[0], mel)
objs = [None]
dist.all_gather_object(objs, dict(rank=0, pack_export_s=1.5))
assert objs[0]["pack_export_s"] == 1.5
t = torch.tensor([4.25], device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)               # bench.py: max over ranks of the timed region
assert float(t) == 4.25
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_OK")
""" % ROOT


def test_rccl_world_size_1_runs_every_collective_of_the_path():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("DF_DIST_SHARE_GPU0", None)
    out = subprocess.run([sys.executable, "-c", SCRIPT], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
