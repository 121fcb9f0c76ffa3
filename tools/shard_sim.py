#!/usr/bin/env python3
"""Development aid (one GPU): what ONE rank of an N-rank pair-sharded run computes per PCG iteration.  Rank 0's shard of the
benchmark's pairs is solved on a 1-rank RCCL communicator with the sharded code path forced (the other ranks' contributions
are simply missing: a smaller but valid problem), and the kernel classes are timed.  The collectives cost nothing here, so the
figures are the COMPUTE side of an N-GPU iteration; the exchange (one all-reduce of [q | Z^T q | p.q], 445 KB, per product)
comes on top.  usage: shard_sim.py [N ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import copy
import torch
torch.cuda.init()
import bench
from robust_cvd_amd import api, sharding, synth
from robust_cvd_amd.ctypes_types import OptParams

worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
full = synth.make_video(300, 384, 224, seed=bench.SEED, extra_offsets=6)
for world in worlds:
    v = copy.copy(full)
    mine = sharding.shard_pairs(full.pairs, full.offsets, world)[0]
    v.pairs, v.offsets, v.loc, v.is_static = sharding.take_pairs(full.pairs, full.offsets, full.loc, full.is_static, mine)
    s = api.Solver(0)
    s.set_options(force_sharded_path=1)
    s.comm_init(0, 1, api.Solver.comm_unique_id())
    p = OptParams.defaults()
    bench.prepare(s, v, p, pair_graph=full.pairs)
    pose0, theta0 = s.get_pose_params().copy(), s.get_xform_params().copy()
    bench.run_iterations(s, p, pose0, theta0, 3)
    s.set_kernel_timing(True)
    done, cg, solves, last = bench.run_iterations(s, p, pose0, theta0, 12)
    kt = s.kernel_times()
    per_it = kt["matvec_pairs"]["avg_ms"] + kt["matvec_finish"]["avg_ms"] + kt["cg_update"]["avg_ms"]
    print(f"world {world}: rank 0 holds {len(v.pairs)} pairs / {int(v.offsets[-1])} constraints; per PCG iteration (HIP events, "
          f"incl. dispatch gaps): product {kt['matvec_pairs']['avg_ms'] * 1e3:.1f} us + finish (+ exchange call) "
          f"{kt['matvec_finish']['avg_ms'] * 1e3:.1f} + update {kt['cg_update']['avg_ms'] * 1e3:.1f} = {per_it * 1e3:.1f} us; "
          f"assembly {kt['evaluate_assemble']['avg_ms']:.3f} ms, preconditioner {kt['block_inverse']['avg_ms']:.3f} ms; "
          f"{cg / done:.1f} PCG iterations per LM iteration", flush=True)
    s.close()
