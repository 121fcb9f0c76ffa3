import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robust_cvd_amd import api, synth
from robust_cvd_amd.ctypes_types import *
v = synth.make_video(2, 96, 56, seed=41, max_pairs=2)
s = api.Solver(0); synth.load_into(s, v)
s.reset_depth_xforms(XformDesc.grid_depth(17, 10)); s.reset_spatial_xforms(XformDesc.spatial())
th = s.get_xform_params(False); th = 0.5 + np.random.default_rng(1).uniform(0, 1.0, th.shape); s.set_xform_params(th, False)
np.savez("gpurun_out/dense_dbg.npz", apply=s.apply_depth_xforms(0, 1), pmap=s.depth_param_maps(0, 1), th=th)
