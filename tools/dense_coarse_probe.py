#!/usr/bin/env python3
"""GPU: run-to-run variation of the dense-mode solve of tests/test_gpu_two_ranks.py's 7-frame problem with the DENSE
coarse level in force (coarse_update_budget = 0: rebuilt at every damping change).  Prints the PCG iterations of every LM
iteration; an entry equal to pcg_max_iterations means the preconditioner was not positive definite there."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robust_cvd_amd import api, synth
from robust_cvd_amd.ctypes_types import OptParams, XformDesc

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
mf = int(sys.argv[2]) if len(sys.argv) > 2 else 0
shift = float(sys.argv[3]) if len(sys.argv) > 3 else None
v = synth.make_video(7, 96, 56, seed=55)
flow, mask = synth.make_dense_flows(v)
for r in range(runs):
    s = api.Solver(0)
    s.set_options(dense_matrix_free=mf, coarse_update_budget=0, coarse_over_budget=1)
    if shift is not None:
        s.set_options(coarse_dense_shift=shift)
    s.set_video(v.num_frames, v.width, v.height, v.aspect, v.inv_aspect)
    s.set_depth_all(v.depth)
    s.reset_poses()
    s.set_pair_flows(v.pairs, flow, mask)
    s.reset_depth_xforms(XformDesc.global_depth())
    s.reset_spatial_xforms(XformDesc.spatial())
    p = OptParams.defaults()
    p.ctf_long, p.ctf_short = 6, 4
    s.normalize_depth(p)
    s.pose_optimization(p)
    sm = s.summary()
    its = [rec["linear_iterations"] for rec in s.records()]
    print(f"run {r}: cost {sm['final_cost']:.12f} LM {sm['num_iterations']} PCG {sm['total_linear_iterations']} capped {sum(1 for x in its if x >= 300)}  {its}", flush=True)
    s.close()
