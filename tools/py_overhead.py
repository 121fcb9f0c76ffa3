import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
import torch; torch.cuda.init()
import bench
from robust_cvd_amd import api, synth
from robust_cvd_amd.ctypes_types import OptParams
v = synth.make_video(300, 384, 224, seed=bench.SEED, extra_offsets=6)
s = api.Solver(0); p = OptParams.defaults(); bench.prepare(s, v, p)
pose0, theta0 = s.get_pose_params().copy(), s.get_xform_params().copy()
for r in range(4):
    t0 = time.perf_counter(); s.set_pose_params(pose0); t1 = time.perf_counter(); s.set_xform_params(theta0); t2 = time.perf_counter()
    p.max_iterations = 4
    s.pose_optimization_step(p, p.depth_deform_reg_final, convert_poses=False); t3 = time.perf_counter()
    sm = s.summary(); t4 = time.perf_counter()
    print("set_pose %.0f us  set_xform %.0f us  solve %.0f us (library total_seconds %.0f us)  summary %.0f us" % ((t1-t0)*1e6, (t2-t1)*1e6, (t3-t2)*1e6, sm["total_seconds"]*1e6, (t4-t3)*1e6))
