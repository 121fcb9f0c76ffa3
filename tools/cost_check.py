"""Development aid (GPU): candidate-cost kernels vs the assembly kernels at the same (accepted) point."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robust_cvd_amd import api, synth
from robust_cvd_amd.ctypes_types import *
v = synth.make_video(5, 64, 40, seed=43, spacing=9)
for generic in (False, True):
    for regs in (True, False):
        s = api.Solver(0); synth.load_into(s, v)
        s.reset_depth_xforms(XformDesc.grid_depth(5, 4)); s.reset_spatial_xforms(XformDesc.spatial())
        rng = np.random.default_rng(11); F = v.num_frames
        pose = np.zeros((F, 7)); pose[:, :6] = rng.normal(0, 0.03, (F, 6)); pose[:, 6] = 0.2 + rng.uniform(0, 0.02, F)
        dx = 0.15 + rng.uniform(0, 0.05, s.get_xform_params().shape); s.set_xform_params(dx); s.set_pose_params(pose)
        s.set_options(pcg_relative_tolerance=1e-10, coarse_level=0); s.set_generic_kernels(generic)
        p = OptParams.defaults(); p.max_iterations = 3
        reg = 0.1
        if not regs:
            p.scale_reg = 0.0; p.focal_reg = 0.0; reg = 0.0
        s.pose_optimization_step(p, reg, convert_poses=False)
        sm = s.summary()
        again = s.evaluate(p, reg, want_gradient=False)["cost"]
        again2 = s.evaluate(p, reg, s.get_pose_params(), want_gradient=False)["cost"]
        print(f"generic={generic} regs={regs}: final {sm['final_cost']:.15e} evaluate {again:.15e} (explicit pose {again2:.15e}) rel {abs(again-sm['final_cost'])/again:.2e} steps {sm['num_successful_steps']}")
