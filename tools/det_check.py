#!/usr/bin/env python3
"""GPU development aid: repeat the benchmark's timed solve from the same start state and print the PCG iterations of
every LM iteration -- the sequence must not depend on timing.  usage: det_check.py [pairs_level] [repeats]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import bench
from robust_cvd_amd import api, synth
from robust_cvd_amd.ctypes_types import OptParams

level = int(sys.argv[1]) if len(sys.argv) > 1 else 6
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
v = synth.make_video(300, 384, 224, seed=bench.SEED, extra_offsets=level)
s = api.Solver(0)
p = OptParams.defaults()
bench.prepare(s, v, p)
pose0, theta0 = s.get_pose_params().copy(), s.get_xform_params().copy()
if os.environ.get("DET_BENCH"):  # the benchmark's own sequence: 3 warm-up iterations, then solves until 20 iterations are done
    print("prepare", s.summary()["total_linear_iterations"], "%.12g" % s.summary()["final_cost"])
    for count in (3, 20):
        done = 0
        while done < count:
            s.set_pose_params(pose0)
            s.set_xform_params(theta0)
            p.max_iterations = count - done
            s.pose_optimization_step(p, p.depth_deform_reg_final, convert_poses=False)
            recs = s.records()
            done += s.summary()["num_iterations"]
            print(count, [int(x["linear_iterations"]) for x in recs[1:]], "%.12g" % s.summary()["final_cost"], flush=True)
            if recs[-1]["linear_iterations"] > 100 or os.environ.get("DET_ALL"):
                for x in recs:
                    print("    ", {k: (("%.6e" % v) if isinstance(v, float) else v) for k, v in x.items()})
    sys.exit(0)
for r in range(reps):
    s.set_pose_params(pose0)
    s.set_xform_params(theta0)
    p.max_iterations = 8
    s.pose_optimization_step(p, p.depth_deform_reg_final, convert_poses=False)
    recs = s.records()
    print(r, [int(x["linear_iterations"]) for x in recs[1:]], "%.12g" % s.summary()["final_cost"], flush=True)
