#!/usr/bin/env python3
"""Development aid (one GPU): what ONE rank of an N-rank pair-sharded run computes per PCG iteration.  The handle becomes rank 0
of a PHANTOM N-rank run (cvd_comm_init_phantom: the other ranks do not exist, every collective returns at once): it holds rank
0's shard of the benchmark's pairs, owns rank 0's chunk of the frames and runs the sharded code path with that rank's real launch
geometry -- product over its pairs, finish over all frames, update over ITS frames (owner-sharded iteration) -- while the kernel
classes are timed.  The collectives cost nothing here, so the figures are the COMPUTE side of an N-GPU iteration; the numbers
the solve produces mean nothing (the other ranks' contributions are missing), which is why the iteration counts are forced.
usage: shard_sim.py [N ...] [--replicated] [--config4] [--dense] [--frames F]
  --replicated: cvd_solver_options::dist_owner_update = 0, the round-3 scheme
  --config4:    BASELINE configs[4] (1000 frames 640 x 384, 16 x 12 grid, 5958 pairs) instead of configs[2] with 4140 pairs
  --dense:      configs[2] in dense mode (every masked pixel of 1766 pairs): the per-rank figure that matters there is the ASSEMBLY
                (the pixel walk of a Jacobian evaluation is per pair and divides by the ranks)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import copy
import torch
torch.cuda.init()
import bench
from robust_cvd_amd import api, sharding, synth
from robust_cvd_amd.ctypes_types import OptParams

replicated = "--replicated" in sys.argv
config4 = "--config4" in sys.argv
dense = "--dense" in sys.argv
argv = [a for a in sys.argv[1:]]
frames_override = None
if "--frames" in argv:
    k = argv.index("--frames")
    frames_override = int(argv[k + 1])
    del argv[k:k + 2]
worlds = [int(a) for a in argv if not a.startswith("--")] or [1, 2, 4, 8]
p = OptParams.defaults()
if config4:
    cfg = bench.CONFIGS[4]
    p.ctf_long, p.ctf_short = cfg["ctf"]
    full = synth.make_video(frames_override or cfg["frames"], cfg["width"], cfg["height"], seed=bench.SEED, extra_offsets=1)
elif dense:
    full = synth.make_video(frames_override or 300, 384, 224, seed=bench.SEED, extra_offsets=1, spacing=1e9)
    full.dense_flow, full.dense_mask = synth.make_dense_flows(full)
else:
    full = synth.make_video(frames_override or 300, 384, 224, seed=bench.SEED, extra_offsets=6)
# the state the timed iterations start from, from a REAL single-rank run (the phantom solve below cannot produce one)
ref = api.Solver(0)
bench.prepare(ref, full, p)
pose0, theta0 = ref.get_pose_params().copy(), ref.get_xform_params().copy()
desc = ref.xform_desc()
ref.close()
for world in worlds:
    v = copy.copy(full)
    if dense:
        mine = sharding.shard_pairs(full.pairs, sharding.uniform_offsets(len(full.pairs), full.width * full.height), world)[0]
        _, v.dense_flow, v.dense_mask = sharding.take_pair_flows(full.pairs, full.dense_flow, full.dense_mask, mine)
    else:
        mine = sharding.shard_pairs(full.pairs, full.offsets, world)[0]
    v.pairs, v.offsets, v.loc, v.is_static = sharding.take_pairs(full.pairs, full.offsets, full.loc, full.is_static, mine)
    s = api.Solver(0)
    s.comm_init_phantom(0, world)
    # fixed work: 3 LM iterations of exactly 40 PCG iterations each (nothing converges with the other ranks missing)
    s.set_options(force_sharded_path=1, dist_owner_update=int(not replicated), pcg_max_iterations=40, pcg_relative_tolerance=1e-12,
                  force_iterations=1)
    synth.load_into(s, v, p.focal_long)
    if dense:
        s.set_pair_flows(v.pairs, v.dense_flow, v.dense_mask)
    s.set_pair_graph(full.pairs)
    from robust_cvd_amd.ctypes_types import XformDesc
    s.reset_depth_xforms(desc)
    s.reset_spatial_xforms(XformDesc.spatial())

    def run(n):
        s.set_pose_params(pose0)
        s.set_xform_params(theta0)
        p.max_iterations = n
        s.pose_optimization_step(p, p.depth_deform_reg_final, convert_poses=False)

    run(2)
    s.set_kernel_timing(True)
    run(3)
    kt = s.kernel_times()
    dk = s.dense_times() if dense else None
    sm = s.summary()
    per_it = kt["matvec_pairs"]["avg_ms"] + kt["matvec_finish"]["avg_ms"] + kt["cg_update"]["avg_ms"]
    print(f"world {world}{' (replicated update)' if replicated else ''}{' configs[4]' if config4 else (' dense' if dense else '')}: rank 0 holds {len(v.pairs)} pairs / {int(s.num_active_constraints())} "
          f"constraints and owns {-(-full.num_frames // world)} frames; per PCG iteration (HIP events, incl. dispatch gaps): product "
          f"{kt['matvec_pairs']['avg_ms'] * 1e3:.1f} us + finish {kt['matvec_finish']['avg_ms'] * 1e3:.1f} + update "
          f"{kt['cg_update']['avg_ms'] * 1e3:.1f} = {per_it * 1e3:.1f} us; assembly {kt['evaluate_assemble']['avg_ms']:.3f} ms, "
          f"preconditioner {kt['block_inverse']['avg_ms']:.3f} ms; {sm['total_linear_iterations']} PCG iterations in "
          f"{sm['num_iterations']} LM iterations"
          + (f"; dense walk {dk['dense_walk']['avg_ms']:.3f} ms + grid x grid {dk['dense_gg']['avg_ms']:.3f} ms per Jacobian evaluation" if dense else ""),
          flush=True)
    s.close()
