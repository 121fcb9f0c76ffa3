"""Development aid (GPU): run the coarse-to-fine pipeline on a small video, stop the last level early, dump the state."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robust_cvd_amd import api, synth
from robust_cvd_amd.ctypes_types import OptParams, XformDesc

F = int(sys.argv[1]); its = int(sys.argv[2]); out = sys.argv[3]
v = synth.make_video(F, 192, 112, seed=1237)
s = api.Solver(0); synth.load_into(s, v)
s.set_options(verbose=1)
s.reset_depth_xforms(XformDesc.global_depth()); s.reset_spatial_xforms(XformDesc.spatial())
p = OptParams.defaults()
s.normalize_depth(p)
s.pose_optimization_step(p, 0.1)
for gx, gy in ((6, 4), (12, 7), (17, 10)):
    s.grid_xform_split(XformDesc.grid_depth(gx, gy))
    if gx == 17:
        p.max_iterations = its
    s.pose_optimization_step(p, 0.1)
rec = s.records()
np.savez(out, pose=s.get_pose_params(), theta=s.get_xform_params(False), radius=rec[-1]["trust_region_radius"], F=F)
print("saved", out, rec[-1])
