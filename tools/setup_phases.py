#!/usr/bin/env python3
"""GPU: where the fixed cost of one solve goes on the host (cvd_solver_options::verbose = 3 prints the set-up phases of solve())
and around it (Python / ctypes: state hand-over).  The benchmark's timed solves are 3 LM iterations long, so this cost is paid every
third iteration.  usage: setup_phases.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import bench
from robust_cvd_amd import api, synth
from robust_cvd_amd.ctypes_types import OptParams

v = synth.make_video(300, 384, 224, seed=bench.SEED, extra_offsets=6)
s = api.Solver(0)
p = OptParams.defaults()
bench.prepare(s, v, p)
pose0, theta0 = s.get_pose_params().copy(), s.get_xform_params().copy()
for rep in range(3):
    t0 = time.perf_counter()
    s.set_pose_params(pose0)
    s.set_xform_params(theta0)
    t1 = time.perf_counter()
    p.max_iterations = 3
    if rep == 2:
        s.set_options(verbose=3)
    s.pose_optimization_step(p, p.depth_deform_reg_final, convert_poses=False)
    t2 = time.perf_counter()
    sm = s.summary()
    t3 = time.perf_counter()
    print(f"rep {rep}: hand-over {1e6 * (t1 - t0):.0f} us, solve call {1e6 * (t2 - t1):.0f} us (library total_seconds {1e6 * sm['total_seconds']:.0f} us, "
          f"evaluate {1e6 * sm['evaluate_seconds']:.0f}, linear {1e6 * sm['linear_solve_seconds']:.0f}), summary {1e6 * (t3 - t2):.0f} us, LM {sm['num_iterations']}", flush=True)
