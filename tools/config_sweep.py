"""Development aid (GPU): run the BASELINE.json configurations end to end and print LM / PCG counts and times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robust_cvd_amd import api, synth
from robust_cvd_amd.ctypes_types import *

_videos = {}
def run(name, F, W, H, setup, params_mod=None, steps=None, robust=0, seed=None):
    key = (F, W, H, seed)
    if key not in _videos: _videos[key] = synth.make_video(F, W, H, seed=seed if seed is not None else 1234 + len(name))
    v = _videos[key]
    s = api.Solver(0); synth.load_into(s, v)
    s.set_robust_loss(robust)
    p = OptParams.defaults()
    if params_mod: params_mod(p)
    t0 = time.time()
    setup(s, p)
    dt = time.time() - t0
    sm = s.summary()
    err = synth.relative_pose_error(s.get_poses()["position"], s.get_poses()["orientation"], v.true_t, None) if False else None
    print(f"{name}: {F} frames {W}x{H}, {v.num_constraints} constraints: {dt*1e3:.1f} ms, LM {sm['num_iterations']}, "
          f"PCG {sm['total_linear_iterations']}, cost {sm['initial_cost']:.4g} -> {sm['final_cost']:.6g}, term {sm['termination']}")

def cfg0(s, p):  # 30-frame 192x112, fixed intrinsics, global-scale-only deformation
    s.reset_depth_xforms(XformDesc.global_depth()); s.reset_spatial_xforms(XformDesc.spatial())
    s.normalize_depth(p); s.pose_optimization_step(p, 0.1)
def mod0(p): p.intr_opt = IntrinsicsOptimization.Fixed; p.coarse_to_fine = 0
def cfg1(s, p):  # 100-frame 384x224, 4x4 bicubic grid
    s.reset_depth_xforms(XformDesc.global_depth()); s.reset_spatial_xforms(XformDesc.spatial())
    s.normalize_depth(p); s.pose_optimization_step(p, 0.1)
    s.reset_depth_xforms(XformDesc.grid_depth(4, 4, cubic=True)); s.pose_optimization_step(p, 0.1)
def cfg2(s, p):  # 300-frame full pipeline
    s.reset_depth_xforms(XformDesc.global_depth()); s.reset_spatial_xforms(XformDesc.spatial())
    s.normalize_depth(p); s.pose_optimization(p)
def cfg4(s, p):  # 16x12 grid (B = 199), larger frames
    s.reset_depth_xforms(XformDesc.global_depth()); s.reset_spatial_xforms(XformDesc.spatial())
    s.normalize_depth(p); s.pose_optimization(p)
def mod4(p): p.ctf_long, p.ctf_short = 16, 12

run("configs[0]", 30, 192, 112, cfg0, mod0)
run("configs[1]", 100, 384, 224, cfg1)
run("configs[2]", 300, 384, 224, cfg2)
# configs[4] on ONE GPU (the 8-GPU run is the driver's): 1000 frames 640x384, 16x12 grid, the reference's Cauchy loss and
# the Huber stress variant BASELINE.json names (cvd_solver_options::robust_loss = 1)
F4 = int(os.environ.get("CFG4_FRAMES", "1000"))
run("configs[4] Cauchy 0.5", F4, 640, 384, cfg4, mod4, seed=1238)
run("configs[4] Huber 0.5", F4, 640, 384, cfg4, mod4, robust=1, seed=1238)
def mod4h(p): mod4(p); p.robustness = 0.01
run("configs[4] Huber 0.01", F4, 640, 384, cfg4, mod4h, robust=1, seed=1238)
