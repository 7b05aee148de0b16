"""CPU (gloo, world_size 2): the multi-GPU plumbing -- sharding by global sample index and the single flat
weight broadcast -- without a GPU."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import diff_foley_amd  # noqa: F401
from diff_foley_amd import parallel, synth


def test_shard_range_partitions_exactly():
    for G in (1, 4, 7, 64):
        for W in (1, 2, 3, 8):
            spans = [parallel.shard_range(G, r, W) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == G
            assert all(spans[i][1] == spans[i + 1][0] for i in range(W - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_inputs_are_rank_count_invariant():
    """x_T / features of global sample i do not depend on how the batch is sharded."""
    full = synth.synthetic_xT(8)
    for W in (2, 4):
        parts = []
        for r in range(W):
            lo, hi = parallel.shard_range(8, r, W)
            parts.append(synth.synthetic_xT(hi - lo, first_index=lo))
        assert torch.equal(torch.cat(parts), full)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_process_group("gloo")
    spec = synth.state_dict_spec(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY)
    sd = synth.make_state_dict(spec, 3) if r == 0 else None
    got = parallel.broadcast_state_dict(sd, spec, torch.device("cpu"), src=0)
    ref = synth.make_state_dict(spec, 3)
    ok = all(torch.equal(got[k], ref[k]) for k in spec) and list(got.keys()) == list(spec.keys())
    lo, hi = parallel.shard_range(6, r, w)
    mine = torch.arange(lo, hi, dtype=torch.float32).reshape(-1, 1)
    allv = parallel.gather_to_rank0(mine)
    if r == 0:
        ok = ok and torch.equal(allv.flatten(), torch.arange(6, dtype=torch.float32))
    lo, hi = parallel.shard_range(5, r, w)                    # ragged: 3 + 2 samples
    allv = parallel.gather_to_rank0(torch.arange(lo, hi, dtype=torch.float32).reshape(-1, 1, 1).repeat(1, 2, 3))
    if r == 0:
        ok = ok and allv.shape == (5, 2, 3) and torch.equal(allv[:, 0, 0], torch.arange(5, dtype=torch.float32))
    payload = torch.arange(1000, dtype=torch.int64).to(torch.uint8) if r == 0 else None   # only src knows the length
    got_b = parallel.broadcast_bytes(payload, 0, torch.device("cpu"))
    ok = ok and got_b.numel() == 1000 and int(got_b[999]) == 999 % 256
    q.put((r, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_weight_broadcast_and_gather_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
    assert res == {0: True, 1: True}
