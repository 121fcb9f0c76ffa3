#!/usr/bin/env python3
"""GPU: run-to-run spread of the quantities tests/test_reference_reprojection.py asserts (margin policy: an assert sits >= 3 x above
the worst value of 5 repeated runs).  usage: reproj_repeat.py [runs]"""
import importlib, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robust_cvd_amd import build as b, synth
from tests import reference_reprojection as rr
d = os.path.dirname(b.build_lib_python())
sys.path.insert(0, d)
lib = importlib.import_module("lib_python")
g = dict(np.load(rr.GOLDEN))
video = rr.make_case()
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
first = None
for r in range(runs):
    with tempfile.TemporaryDirectory() as tmp:
        out = rr.run_drop_in(lib, video, os.path.join(tmp, "video"))
    perr, rerr = synth.relative_pose_error(out["position"], out["orientation"], g["position"], g["orientation"])
    gs, gg = np.median(out["params"]), np.median(g["params"])
    shape = np.abs(out["params"] / gs / (g["params"] / gg) - 1.0).max()
    mshape = np.abs(out["param_map"][g["map_frames"]] / gs / (g["param_map"] / gg) - 1.0).max()
    ext, intr = rr.numpy_update_poses(out)
    fa, fb, pix, target, depth = rr.constraint_samples(video, out)
    err = np.linalg.norm(rr.numpy_reproject(ext, intr, fa, fb, pix, depth) - target, axis=1)
    spread, _ = rr.depth_scale_spread(video, out)
    if first is None:
        first = out
    print(f"run {r}: pose err vs fixture {perr:.2e} / {rerr:.2e}  fov {np.abs(out['vfov'] - g['vfov']).max():.2e}  gauge {gs / gg - 1.0:+.3e}  "
          f"scale-free params {shape:.2e} map {mshape:.2e}  reproj max {err.max():.4f} mean {err.mean():.5f}  spread {spread:.3e}  "
          f"vs run 0: params {np.abs(out['params'] / first['params'] - 1).max():.2e}", flush=True)
