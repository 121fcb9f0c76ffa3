"""Shared test helpers: load a golden case into any Binding (product Solver or test Oracle)."""
import ctypes as C
import glob
import os

import numpy as np

from robust_cvd_amd.ctypes_types import OptParams, XformDesc

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def desc_from_bytes(arr):
    d = XformDesc()
    C.memmove(C.byref(d), arr.tobytes(), C.sizeof(XformDesc))
    return d


def evaluate_golden(binding, g, **kw):
    binding.set_video(int(g["frames"]), int(g["width"]), int(g["height"]), float(g["aspect"]), float(g["inv_aspect"]))
    binding.set_depth_all(g["depth"])
    binding.set_pair_constraints(g["pairs"], g["offsets"], g["loc"], g["is_static"])
    binding.reset_depth_xforms(desc_from_bytes(g["depth_desc"]))
    binding.reset_spatial_xforms(desc_from_bytes(g["spatial_desc"]))
    binding.set_xform_params(g["depth_params"], False)
    binding.set_xform_params(g["spatial_params"], True)
    p = OptParams.defaults()
    p.num_threads = 1
    p.intr_opt = int(g["intr_opt"])
    p.static_loss_type = int(g["loss"])
    if "smooth" in g:  # scene-flow smoothness triplets
        binding.set_triplet_constraints(g["trip_centers"], g["trip_offsets"], g["trip_loc"], g["trip_static"])
        p.smooth_loss_type = int(g["smooth"][0])
        p.smooth_static_weight = float(g["smooth"][1])
        p.smooth_dynamic_weight = float(g["smooth"][2])
    return binding.evaluate(p, float(g["depth_deform_reg"]), g["pose"], want_gradient=True, want_hdiag=True, **kw)


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))
