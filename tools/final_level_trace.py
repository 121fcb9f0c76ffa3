#!/usr/bin/env python3
"""GPU: per-LM-iteration PCG counts of the final coarse-to-fine level (the iterations bench.py times).
usage: final_level_trace.py [pairs_level] [solves]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()
import bench
from robust_cvd_amd import api, synth
from robust_cvd_amd.ctypes_types import OptParams
level = int(sys.argv[1]) if len(sys.argv) > 1 else 6
solves = int(sys.argv[2]) if len(sys.argv) > 2 else 2
params = OptParams.defaults()
video = synth.make_video(300, 384, 224, seed=bench.SEED, extra_offsets=level)
s = api.Solver(0)
if os.environ.get("ETA"):
    s.set_options(pcg_relative_tolerance=float(os.environ["ETA"]))
bench.prepare(s, video, params)
pose0, theta0 = s.get_pose_params().copy(), s.get_xform_params().copy()
for k in range(solves):
    s.set_pose_params(pose0); s.set_xform_params(theta0)
    params.max_iterations = int(os.environ.get("MAXIT", "1000"))
    t0 = time.perf_counter()
    s.pose_optimization_step(params, params.depth_deform_reg_final, convert_poses=False)
    dt = time.perf_counter() - t0
    sm = s.summary()
    recs = [r for r in s.records()][-(sm["num_iterations"] + 1):]
    print(f"level {level} solve {k}: {sm['num_iterations']} LM it, {sm['total_linear_iterations']} PCG, {dt * 1e3:.2f} ms "
          f"({dt * 1e3 / max(1, sm['num_iterations']):.2f} ms/it), cost {sm['initial_cost']:.6f} -> {sm['final_cost']:.9f}; PCG per it: "
          + " ".join(str(r["linear_iterations"]) for r in recs[1:]))
