#!/usr/bin/env python3
"""GPU: the pair-sharded code path (owner exchange, per-product all-reduce, sparsified coarse level with all-reduced blocks) on
the FULL benchmark problem with a 1-rank RCCL communicator (solver option force_sharded_path), against the plain single-GPU solve of the same
level.  N > 1 needs N GPUs; this is what one box can check.  usage: forced_dist_check.py [pairs_level]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import bench
from robust_cvd_amd import api, synth
from robust_cvd_amd.ctypes_types import OptParams

level = int(sys.argv[1]) if len(sys.argv) > 1 else 6
v = synth.make_video(300, 384, 224, seed=bench.SEED, extra_offsets=level)
res = {}
for mode in ("single", "forced_dist"):
    s = api.Solver(0)
    if mode == "forced_dist":
        s.set_options(force_sharded_path=1)
        s.comm_init(0, 1, api.Solver.comm_unique_id())
    p = OptParams.defaults()
    bench.prepare(s, v, p, pair_graph=(v.pairs if mode == "forced_dist" else None))
    pose0, theta0 = s.get_pose_params().copy(), s.get_xform_params().copy()
    p.max_iterations = 1000
    s.pose_optimization_step(p, p.depth_deform_reg_final, convert_poses=False)   # (warm-up: first launches of this level)
    s.set_pose_params(pose0)
    s.set_xform_params(theta0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    p.max_iterations = 1000
    s.pose_optimization_step(p, p.depth_deform_reg_final, convert_poses=False)
    dt = time.perf_counter() - t0
    sm = s.summary()
    res[mode] = (sm, s.get_poses(), s.get_xform_params().copy())
    print(mode, "PCG per LM iteration", [int(r["linear_iterations"]) for r in s.records()[1:]])
    print(mode, "LM", sm["num_iterations"], "PCG", sm["total_linear_iterations"], "cost %.10f" % sm["final_cost"], "%.1f ms" % (dt * 1e3),
          s.comm_times() if mode == "forced_dist" else "")
a, b = res["single"], res["forced_dist"]
perr, rerr = synth.relative_pose_error(b[1]["position"], b[1]["orientation"], a[1]["position"], a[1]["orientation"])
print("final cost rel diff %.2e  pose err %.2e  rot err %.2e  theta rel diff %.2e" % (
    abs(a[0]["final_cost"] - b[0]["final_cost"]) / abs(a[0]["final_cost"]), perr, rerr,
    float(abs(a[2] - b[2]).max() / abs(a[2]).max())))
