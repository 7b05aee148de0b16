"""GPU: the N > 1 path end to end on ONE GPU (the test box has a single MI355X): two ranks share cuda:0 through the
DF_DIST_SHARE_GPU0 hook of parallel.init_process_group (gloo transport), rank 0 packs the checkpoint once and broadcasts
the packed operand blob, rank 1 imports it (no fp32 checkpoint there), both sample their shard of a global batch of 4
with x_T / features seeded by GLOBAL sample index, and the gathered decoded mels must be BIT-EQUAL to the same shards run
one after the other on one rank (results are rank-count invariant)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

G, STEPS = 4, 6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _sample(model, lo, hi, synth):
    feats = synth.synthetic_cavp(G, 32, 64, seed=1234)[lo:hi].cuda()
    xT = synth.synthetic_xT(hi - lo, first_index=lo).cuda()
    c = model.get_learned_conditioning(feats)
    z, _ = model.sample_log_diff_sampler(c, hi - lo, "DDIM", STEPS, unconditional_guidance_scale=4.5,
                                         unconditional_conditioning=torch.zeros_like(c), x_T=xT)
    return model.decode_first_stage(z)[:, 0]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), DF_DIST_SHARE_GPU0="1",
                      DF_RAW_DATA_MAX="5120")      # tiny model: weight matrices travel shape-only, like the full model's
    import diff_foley_amd as P
    from diff_foley_amd import parallel, synth
    try:
        r, w, local = parallel.init_process_group()
        assert local == 0 and w == world
        cfg = P.stage2_config(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY)
        m = P.LatentDiffusion(**cfg)
        if r == 0:
            spec = synth.state_dict_spec(synth.UNET_TINY, synth.VAE_TINY, synth.COND_TINY)
            m.load_state_dict(synth.make_state_dict(spec, 0))
        m.cuda(0)
        info = parallel.broadcast_packed_model(m, G // world, src=0)
        lo, hi = parallel.shard_range(G, r, w)
        mel = _sample(m, lo, hi, synth)
        allm = parallel.gather_to_rank0(mel.cpu())
        ok, msg = True, ""
        if r == 0:
            # (a) the same shards computed one after the other on ONE rank (same plan shape): bit-identical -- the result
            #     of global sample i does not depend on which rank ran it, nor on the imported-vs-packed-locally weights
            seq = torch.cat([_sample(m, *parallel.shard_range(G, k, w), synth).cpu() for k in range(w)])
            same = torch.equal(allm, seq)
            h = G // w
            d0 = float((allm[:h] - seq[:h]).abs().max())       # rank 0's shard, run twice on rank 0
            d1 = float((allm[h:] - seq[h:]).abs().max())       # rank 1's shard: rank 1 (imported blob) vs rank 0
            # (b) the whole global batch in one plan (other tiles -> other fp32 summation order, amplified by 6 CFG
            #     steps of a random-weight net): sampler-trajectory tolerance of test_path_gpu.py
            ref = _sample(m, 0, G, synth).cpu()
            err = float((allm - ref).norm() / ref.norm())
            # (a) must be BIT-equal.  Round 2 saw ~1 launch in 10^4 differ when two processes shared the GPU: a ring slot of the
            # LDS-DMA GEMM kernels was handed back to the DMA engine while another wave's ds_reads of it were still in flight
            # (fixed in round 3: s_waitcnt lgkmcnt(0) in front of every ring barrier, csrc/gemm_impl.h DF_RING_SYNC;
            # tools/chk_probe.py localised it).
            err_seq = float((allm - seq).norm() / seq.norm())
            ok = allm.shape == ref.shape and same and err < 5e-2
            msg = (f"2 ranks on one GPU: packed blob {info['blob_bytes'] / 1e6:.1f} MB + manifest {info['manifest_bytes'] / 1e3:.1f} KB, "
                   f"pack+export {info['pack_export_s'] * 1e3:.0f} ms, bcast {info['bcast_s'] * 1e3:.0f} ms; gathered mels == "
                   f"sequential shards on one rank: {same} (max|d| own shard rerun {d0:.1e}, other rank's shard {d1:.1e}); vs the global batch in one plan rel-L2 {err:.2e}")
        else:
            # the importing rank never saw an fp32 checkpoint: a plan that needs a packing which was not exported (a 40-frame
            # context takes the K / V^T cross-attention form: separate to_k / to_v / LayerNorm-folded to_q operands) must fail
            # LOUDLY, naming the situation -- not sample from uninitialised operands
            try:
                m.engine.set_context(torch.zeros(2, 40, 128).cuda())
                m.engine.unet_forward(torch.zeros(2, 4, 16, 64).cuda(), torch.zeros(2).cuda())
                ok, msg = False, "an un-exported packing was built on the importing rank without an fp32 checkpoint"
            except RuntimeError as ex:
                ok = "imported shape-only" in str(ex)
                msg = "" if ok else f"unexpected error text: {ex}"
            # ... while another batch size of the EXPORTED shape class needs no new packing and must simply work
            if ok:
                c1 = m.get_learned_conditioning(synth.synthetic_cavp(G, 32, 64, seed=1234)[:1].cuda())
                z1, _ = m.sample_log_diff_sampler(c1, 1, "DDIM", 2, unconditional_guidance_scale=4.5,
                                                  unconditional_conditioning=torch.zeros_like(c1), x_T=synth.synthetic_xT(1).cuda())
                ok = bool(torch.isfinite(z1).all())
                msg = "" if ok else "batch 1 on the importing rank gave non-finite latents"
        q.put((r, bool(ok), msg))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    except Exception as ex:     # noqa: BLE001
        q.put((rank, False, repr(ex)))


def test_two_ranks_share_gpu0_outputs_match_single_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(60)
    for r, ok, msg in sorted(res):
        if msg:
            print(f"rank {r}: {msg}")
    assert all(ok for _, ok, _ in res), res
