"""N > 1 path on CPU: world_size-2 `gloo` run of the pair-sharded evaluation.

Each rank evaluates ITS shard of the frame pairs (with the oracle standing in for the device evaluator) plus
the regularisers of the frames it owns, and one all-reduce (SUM) of [cost | gradient | diagonal blocks]
reproduces the unsharded evaluation -- the decomposition the RCCL path relies on."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle.oracle import Oracle
    from robust_cvd_amd import sharding, synth
    from robust_cvd_amd.ctypes_types import OptParams, XformDesc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    v = synth.make_video(8, 64, 40, seed=41, spacing=9)
    p = OptParams.defaults()
    p.num_threads = 1

    def evaluate(pairs, offsets, loc, st, frames, with_static_regs):
        o = Oracle()
        o.set_video(v.num_frames, v.width, v.height, v.aspect, v.inv_aspect)
        o.set_depth_all(v.depth)
        o.set_pair_constraints(pairs, offsets, loc, st)
        o.reset_poses()
        o.reset_depth_xforms(XformDesc.grid_depth(3, 3))
        o.reset_spatial_xforms(XformDesc.spatial())
        pp = OptParams.defaults()
        pp.num_threads = 1
        if not with_static_regs:
            pp.scale_reg = 0.0
            pp.focal_reg = 0.0
            reg = 0.0
        else:
            reg = 0.1
        if frames is not None:
            pp.set_frame_range(frames)
        return o.evaluate(pp, reg, want_gradient=True, want_hdiag=True)

    shards = sharding.shard_pairs(v.pairs, v.offsets, world)
    mine = sharding.take_pairs(v.pairs, v.offsets, v.loc, v.is_static, shards[rank])
    stat = evaluate(*mine, None, False)                                  # static constraints of my pairs
    empty = sharding.take_pairs(v.pairs, v.offsets, v.loc, v.is_static, np.zeros(0, np.int64))
    owned = [f for f in range(v.num_frames) if f % world == rank]
    regs = evaluate(*empty, owned, True)                                 # regularisers of my frames
    buf = np.concatenate([[stat["cost"] + regs["cost"]], (stat["gradient"] + regs["gradient"]).ravel(),
                          (stat["hdiag"] + regs["hdiag"]).ravel()])
    t = torch.from_numpy(buf.copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if rank == 0:
        full = evaluate(v.pairs, v.offsets, v.loc, v.is_static, None, True)
        ref = np.concatenate([[full["cost"]], full["gradient"].ravel(), full["hdiag"].ravel()])
        q.put((float(np.abs(t.numpy() - ref).max() / np.abs(ref).max()),
               sorted(np.concatenate(shards).tolist()) == list(range(len(v.pairs)))))
    dist.barrier()
    dist.destroy_process_group()


def _dense_worker(rank, world, port, q):
    """Dense mode (matchSeparation = 0): the flow / mask IMAGES shard with the pairs (sharding.take_pair_flows, uniform offsets: every
    pair is width x height pixel slots); a rank's evaluation is that of the constraint list its images stand for."""
    sys.path.insert(0, ROOT)
    import copy
    import torch
    import torch.distributed as dist
    from oracle.oracle import Oracle
    from robust_cvd_amd import sharding, synth
    from robust_cvd_amd.ctypes_types import OptParams, XformDesc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    v = synth.make_video(6, 48, 28, seed=43)
    flow, mask = synth.make_dense_flows(v)

    def evaluate(pairs, fl, mk, frames, with_regs):
        vv = copy.copy(v)
        vv.pairs = pairs
        off, loc = synth.dense_constraints_from_flows(vv, fl, mk) if len(pairs) else (np.zeros(1, np.int64), np.zeros((0, 4), np.float32))
        o = Oracle()
        o.set_video(v.num_frames, v.width, v.height, v.aspect, v.inv_aspect)
        o.set_depth_all(v.depth)
        o.set_pair_constraints(pairs, off, loc, None)
        o.reset_poses()
        o.reset_depth_xforms(XformDesc.grid_depth(3, 3))
        o.reset_spatial_xforms(XformDesc.spatial())
        pp = OptParams.defaults()
        pp.num_threads = 1
        reg = 0.1
        if not with_regs:
            pp.scale_reg = 0.0
            pp.focal_reg = 0.0
            reg = 0.0
        if frames is not None:
            pp.set_frame_range(frames)
        return o.evaluate(pp, reg, want_gradient=True, want_hdiag=True), int(off[-1])

    shards = sharding.shard_pairs(v.pairs, sharding.uniform_offsets(len(v.pairs), v.width * v.height), world)
    mine = sharding.take_pair_flows(v.pairs, flow, mask, shards[rank])
    stat, n_mine = evaluate(*mine, None, False)
    owned = [f for f in range(v.num_frames) if f % world == rank]
    regs, _ = evaluate(v.pairs[:0], flow[:0], mask[:0], owned, True)
    buf = np.concatenate([[stat["cost"] + regs["cost"], float(n_mine)], (stat["gradient"] + regs["gradient"]).ravel(),
                          (stat["hdiag"] + regs["hdiag"]).ravel()])
    t = torch.from_numpy(buf.copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if rank == 0:
        full, n_full = evaluate(v.pairs, flow, mask, None, True)
        ref = np.concatenate([[full["cost"], float(n_full)], full["gradient"].ravel(), full["hdiag"].ravel()])
        loads = [len(sh) for sh in shards]
        q.put((float(np.abs(t.numpy() - ref).max() / np.abs(ref).max()),
               sorted(np.concatenate(shards).tolist()) == list(range(len(v.pairs))) and max(loads) - min(loads) <= 2))
    dist.barrier()
    dist.destroy_process_group()


def test_dense_mode_shards_its_images_with_the_pairs():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dense_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    err, complete = q.get(timeout=240)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert complete
    assert err < 1e-12, err


def test_pair_sharded_evaluation_all_reduces_to_the_full_one():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    err, complete = q.get(timeout=240)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert complete
    assert err < 1e-12, err


def test_shard_pairs_keeps_reverse_pairs_together_and_balances():
    sys.path.insert(0, ROOT)
    from robust_cvd_amd import sharding, synth
    pairs = np.array(synth.hierarchical_pairs(300))
    rng = np.random.default_rng(0)
    counts = rng.integers(500, 700, len(pairs))
    offsets = np.r_[0, np.cumsum(counts)]
    for world in (1, 2, 4, 8):
        shards = sharding.shard_pairs(pairs, offsets, world)
        allidx = np.concatenate(shards)
        assert sorted(allidx.tolist()) == list(range(len(pairs)))
        owner = {}
        for r, sh in enumerate(shards):
            for i in sh:
                owner[tuple(pairs[i])] = r
        assert all(owner[(a, b)] == owner[(b, a)] for a, b in owner)
        loads = np.array([counts[sh].sum() for sh in shards])
        assert loads.max() <= 1.02 * loads.mean() + 1400
