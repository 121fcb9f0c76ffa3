#!/usr/bin/env python3
"""GPU: repeat the benchmark's solve of the final level from the same start state (fresh set_pose_params / set_xform_params every time)
and print, per repeat, the PCG iterations of every LM iteration and the final cost as a hex float.  With the deterministic build
(--variant det: lib/libcvd_hip_det.so, robust_cvd_amd.build.build_deterministic) every line must be identical bit for bit; with the
product build the counts differ by an iteration here and there and the costs in the last digits (LDS atomics of several waves).
usage: det_check.py [--variant det] [--frames 300] [--level 6] [--reps 5] [--iterations 8]"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from robust_cvd_amd import api, synth

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default=None)
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--level", type=int, default=6)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--iterations", type=int, default=8)
ap.add_argument("--pipeline", action="store_true", help="repeat the WHOLE default pipeline (normalizeDepth + coarse-to-fine) on fresh handles instead")
args = ap.parse_args()
if args.variant:
    api.load_library(variant=args.variant)
import bench  # noqa: E402
from robust_cvd_amd.ctypes_types import OptParams  # noqa: E402

v = synth.make_video(args.frames, 384, 224, seed=bench.SEED, extra_offsets=args.level)
if args.pipeline:
    from robust_cvd_amd.ctypes_types import XformDesc
    for r in range(args.reps):
        s = api.Solver(0)
        p = OptParams.defaults()
        synth.load_into(s, v, p.focal_long)
        s.reset_depth_xforms(XformDesc.global_depth())
        s.reset_spatial_xforms(XformDesc.spatial())
        s.normalize_depth(p)
        s.pose_optimization(p)
        sm = s.summary()
        print("rep", r, "LM", sm["num_iterations"], "pcg", sm["total_linear_iterations"], "cost", float(sm["final_cost"]).hex(),
              "pose-digest", hex(hash(s.get_pose_params().tobytes()) & 0xFFFFFFFFFFFF),
              "theta-digest", hex(hash(s.get_xform_params().tobytes()) & 0xFFFFFFFFFFFF), flush=True)
        s.close()
    sys.exit(0)
s = api.Solver(0)
p = OptParams.defaults()
bench.prepare(s, v, p)
pose0, theta0 = s.get_pose_params().copy(), s.get_xform_params().copy()
print("path", s.path_info(), flush=True)
for r in range(args.reps):
    s.set_pose_params(pose0)
    s.set_xform_params(theta0)
    p.max_iterations = args.iterations
    s.pose_optimization_step(p, p.depth_deform_reg_final, convert_poses=False)
    recs = s.records()
    sm = s.summary()
    x = s.get_pose_params()
    print("rep", r, "pcg", [int(q["linear_iterations"]) for q in recs[1:]], "cost", float(sm["final_cost"]).hex(),
          "pose-digest", hex(hash(x.tobytes()) & 0xFFFFFFFFFFFF), flush=True)
