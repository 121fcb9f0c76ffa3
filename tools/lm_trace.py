"""Development aid: print the LM table (incl. PCG iterations per LM iteration) of the full pipeline."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robust_cvd_amd import api, synth
from robust_cvd_amd.ctypes_types import *
F = int(sys.argv[1]) if len(sys.argv) > 1 else 300
tol = float(sys.argv[2]) if len(sys.argv) > 2 else None  # None: the library's default forcing value
v = synth.make_video(F, 384, 224, seed=1237, extra_offsets=int(os.environ.get('CVD_PAIRS_LEVEL', '1')))
s = api.Solver(0); synth.load_into(s, v)
s.set_options(verbose=int(os.environ.get('CVD_VERBOSE', '1')), pcg_relative_tolerance=tol, coarse_level=int(os.environ.get('CVD_COARSE', '1')))
print(len(v.pairs), 'pairs', v.num_constraints, 'constraints')
s.reset_depth_xforms(XformDesc.global_depth()); s.reset_spatial_xforms(XformDesc.spatial())
p = OptParams.defaults()
s.normalize_depth(p)
t0 = time.time(); s.pose_optimization(p); dt = time.time() - t0
sm = s.summary()
print("TOTAL", dt, sm)
if os.environ.get("CVD_TWICE"):
    synth.load_into(s, v)
    s.reset_depth_xforms(XformDesc.global_depth()); s.reset_spatial_xforms(XformDesc.spatial())
    s.normalize_depth(p)
    t0 = time.time(); s.pose_optimization(p); dt = time.time() - t0
    print("SECOND", dt, s.summary()["final_cost"])
