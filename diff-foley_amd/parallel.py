"""Multi-GPU plumbing: one process per GPU, batch-of-videos sharding, ONE weight broadcast at init.

Every latent sample is independent for its whole trajectory, so the global batch is split
contiguously over ranks and the step loop contains no collective.  The only exchange is the initial broadcast
of the fp32 checkpoint from rank 0 as a single flat buffer (RCCL over xGMI when the backend is "nccl";
"gloo" on CPU for the tests), after which each rank re-packs its own bf16 copy.
"""
import os
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist


def init_process_group(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun style).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("DF_DIST_SHARE_GPU0"):          # test hook: several ranks on ONE GPU (1-GPU box), gloo transport
        local = 0
        backend = backend or "gloo"
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


def shard_range(global_batch, rank, world):
    """Contiguous split; the first (global_batch % world) ranks get one extra sample."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_state_dict(state_dict, spec, device, src=0):
    """Broadcast a checkpoint as ONE flat fp32 buffer.

    ``state_dict`` is needed on ``src`` only; ``spec`` (name -> shape, identical on all ranks) fixes the layout.
    Returns an OrderedDict of views into the received buffer (on ``device``)."""
    total = sum(int(np.prod(s)) for s in spec.values())
    flat = torch.empty(total, dtype=torch.float32, device=device)
    rank = dist.get_rank() if dist.is_initialized() else 0
    if rank == src:
        off = 0
        for k, s in spec.items():
            n = int(np.prod(s))
            flat[off:off + n].copy_(state_dict[k].reshape(-1).to(torch.float32), non_blocking=True)
            off += n
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src=src)
    out, off = OrderedDict(), 0
    for k, s in spec.items():
        n = int(np.prod(s))
        out[k] = flat[off:off + n].view(*s)
        off += n
    return out


def gather_to_rank0(t, dst=0):
    """Concatenate per-rank result tensors along dim 0 on rank dst; returns None elsewhere.  Shards may differ in
    their leading dimension (``shard_range`` gives the first G % world ranks one extra sample): the sizes are exchanged
    first and the payload travels padded to the largest shard."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return t
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    nmax = max(sizes)
    pad = t if t.shape[0] == nmax else torch.cat([t, t.new_zeros((nmax - t.shape[0],) + tuple(t.shape[1:]))])
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad.contiguous(), bufs, dst=dst)
    return torch.cat([b[:k] for b, k in zip(bufs, sizes)]) if bufs is not None else None
