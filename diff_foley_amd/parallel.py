"""Multi-GPU plumbing: one process per GPU, batch-of-videos sharding, ONE weight broadcast at init.

Every latent sample is independent for its whole trajectory, so the global batch is split
contiguously over ranks and the step loop contains no collective.  The only exchange is at init: rank 0 packs the
checkpoint ONCE into the engine's MFMA operand layouts and broadcasts that blob (``broadcast_packed_model``: 2.15 GB of
operand-type weights + the small fp32 tensors, RCCL over xGMI when the backend is "nccl"); the other ranks import it --
no fp32 master copies and no re-packing there.  ``broadcast_state_dict`` (the flat fp32 checkpoint, 3.8 GB) remains for
tensors that are not part of a LatentDiffusion (e.g. the classifier) and for the gloo CPU tests.
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


def broadcast_bytes(buf, src, device):
    """Broadcast a uint8 tensor whose length only ``src`` knows (two collectives: size, payload)."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return buf
    n = torch.tensor([buf.numel() if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src)
    out = buf.to(device) if rank == src else torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(out, src=src)
    return out


def broadcast_packed_model(model, batch_size, src=0, size_len=64, context_frames=32):
    """``model``: a LatentDiffusion that is on its device on every rank and holds the checkpoint on ``src`` only.
    ``src`` packs for (batch_size, 16 x size_len latent, context_frames) and exports; everyone else imports.

    Travels in ONE extra byte payload next to the packed blob ("sidecar"): the schedule buffers of ``src`` (a checkpoint may
    carry its own betas / alphas_cumprod: without them the importing ranks would sample with the config's schedule) and the
    autotuner's choices of ``src`` (when its engine tuned the exported plans): every rank then runs the SAME tiles and
    split-K factors -- identical fp32 summation order, bit-equal results across ranks -- and only ``src`` pays a tuning pass.
    The imported choices are applied to every plan an importing rank builds, whether or not it called ``model.autotune()``
    (``df_tune_cache_import``); they are keyed by GEMM shape, so the guarantee covers ranks with EQUAL shard sizes -- a ragged
    last shard (``shard_range``) has its own M, finds no entry and runs the cost model's tiles (or its own tuning pass).
    Every rank must have called ``model.cuda(device)`` before.  Returns seconds spent in (pack + export, broadcast, import) on this rank and the payload sizes."""
    import io
    import time
    from .schedule import BUFFER_NAMES
    rank = dist.get_rank() if dist.is_initialized() else 0
    if model.engine is None:
        raise RuntimeError("broadcast_packed_model: call model.cuda(device) on every rank first")
    dev = model.device
    t0 = time.perf_counter()
    manifest = blob = side = None
    if rank == src:
        manifest, blob = model.export_packed(batch_size, size_len, context_frames)
        bio = io.BytesIO()
        np.savez(bio, tune=np.frombuffer(model.engine.tune_cache_export(), dtype=np.uint8),
                 **{k: getattr(model, k).detach().cpu().numpy() for k in BUFFER_NAMES})
        side = torch.frombuffer(bytearray(bio.getvalue()), dtype=torch.uint8)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    manifest = broadcast_bytes(manifest, src, dev)
    side = broadcast_bytes(side, src, dev)
    blob = broadcast_bytes(blob, src, dev)
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    ntune = 0
    if rank != src:
        z = np.load(io.BytesIO(side.cpu().numpy().tobytes()))
        tune = z["tune"].tobytes()
        ntune = tune.count(b"\n")
        if ntune:
            model.engine.tune_cache_import(tune)
        model.load_packed(manifest, blob)
        for k in BUFFER_NAMES:
            setattr(model, k, torch.from_numpy(z[k]).to(getattr(model, k).device))
    torch.cuda.synchronize(dev)
    return dict(pack_export_s=t1 - t0, bcast_s=t2 - t1, import_s=time.perf_counter() - t2,
                blob_bytes=int(blob.numel()), manifest_bytes=int(manifest.numel()), sidecar_bytes=int(side.numel()),
                tune_entries_imported=ntune, backend=dist.get_backend() if dist.is_initialized() else None)
