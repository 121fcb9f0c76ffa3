#!/usr/bin/env python3
"""Mint the golden vectors in tests/golden/*.npz from the CPU oracle (run from the repo root).

The reference ships no golden vectors and cannot be built here, so these are minted by
oracle/cvd_oracle.cpp after it has been pinned by tests/test_oracle_kat.py and tests/test_oracle_problem.py
(independent numpy restatement, finite differences, scipy).  Each file stores the exact inputs (float32 depth,
constraints, state) and the oracle outputs (cost, gradient, frame-diagonal J^T J blocks), so that both the
oracle (CPU test) and the HIP path (GPU test) are compared against committed numbers.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.oracle import Oracle  # noqa: E402
from robust_cvd_amd import synth  # noqa: E402
from robust_cvd_amd.ctypes_types import OptParams, SpatialXformType, ValueXformType, XformDesc  # noqa: E402

CASES = {
    # name: (frames, w, h, seed, depth desc args, spatial desc args, intr_opt, loss, depth_deform_reg)
    "global_perframe": (5, 64, 40, 21, ("global", 1), ("identity",), 2, 1, 0.1),
    "grid17x10_linear": (4, 96, 56, 22, ("grid", 17, 10, 1, False), ("identity",), 2, 1, 0.1),
    "grid4x4_cubic": (4, 64, 40, 23, ("grid", 4, 4, 1, True), ("identity",), 2, 1, 0.1),
    "grid5x4_cubic_ss_bicubic_spatial": (4, 64, 40, 24, ("grid", 5, 4, 2, True), ("bicubic", 4, 3), 2, 1, 0.1),
    "global_fixed_euclidean": (4, 64, 40, 25, ("global", 2), ("corners",), 0, 0, 0.1),
    "grid3x3_ratio": (4, 64, 40, 26, ("grid", 3, 3, 1, False), ("vertical",), 2, 2, 0.3),
    "grid3x3_log": (4, 64, 40, 27, ("grid", 3, 3, 1, False), ("bilinear", 3, 2), 2, 3, 0.3),
    "grid4x3_shared_intrinsics": (5, 64, 40, 28, ("grid", 4, 3, 1, False), ("identity",), 1, 1, 0.1),
    "global_shared_intrinsics": (5, 64, 40, 29, ("global", 1), ("identity",), 1, 1, 0.1),
    # scene-flow smoothness triplets (a9): name -> same tuple + (smooth loss type, static weight, dynamic weight)
    "grid4x3_triplets_disparity_laplacian": (6, 64, 40, 30, ("grid", 4, 3, 1, False), ("identity",), 2, 1, 0.1, (1, 2.0, 0.5)),
    "global_triplets_euclidean_laplacian": (6, 64, 40, 31, ("global", 1), ("bilinear", 3, 2), 2, 1, 0.1, (0, 1.5, 1.0)),
}


def make_descs(dargs, sargs):
    if dargs[0] == "global":
        dd = XformDesc.global_depth(ValueXformType(dargs[1]))
    else:
        dd = XformDesc.grid_depth(dargs[1], dargs[2], ValueXformType(dargs[3]), cubic=dargs[4])
    kind = {"identity": SpatialXformType.Identity, "bicubic": SpatialXformType.BicubicGrid,
            "bilinear": SpatialXformType.BilinearGrid, "corners": SpatialXformType.CornersBilinear,
            "vertical": SpatialXformType.VerticalLinear}[sargs[0]]
    sd = XformDesc.spatial(kind, *(sargs[1:] if len(sargs) > 1 else (0, 0)))
    return dd, sd


def case_inputs(name):
    F, W, H, seed, dargs, sargs, intr, loss, reg = CASES[name][:9]
    smooth = CASES[name][9] if len(CASES[name]) > 9 else None
    v = synth.make_video(F, W, H, seed=seed, spacing=9)
    rng = np.random.default_rng(seed)
    dd, sd = make_descs(dargs, sargs)
    o = Oracle()
    synth.load_into(o, v)
    o.reset_depth_xforms(dd)
    o.reset_spatial_xforms(sd)
    pose = np.zeros((F, 7))
    pose[:, :3] = rng.normal(0, 0.05, (F, 3))
    pose[:, 3:6] = rng.normal(0, 0.05, (F, 3))
    pose[0, 3:6] = 0
    pose[:, 6] = 0.2 + rng.uniform(0, 0.05, F)
    dx = o.get_xform_params(False)
    if dx.size:
        dx = 0.15 + rng.uniform(0, 0.05, dx.shape)
        if dd.value_xform == 2:
            dx[:, 1::2] = rng.uniform(0, 0.3, dx[:, 1::2].shape)
    sx = rng.normal(0, 0.01, o.get_xform_params(True).shape)
    is_static = (rng.uniform(size=v.num_constraints) > 0.1).astype(np.uint8)  # some dynamic constraints
    depth = v.depth.copy()
    depth[:, ::7, ::5] = 0.0   # invalid depth pixels (quirk q5: such constraints are skipped)
    trip = synth.make_triplets(v, spacing=9.0, seed=seed + 100) if smooth else None
    return v, depth, is_static, pose, dx, sx, dd, sd, intr, loss, reg, smooth, trip


def run_oracle(name):
    v, depth, is_static, pose, dx, sx, dd, sd, intr, loss, reg, smooth, trip = case_inputs(name)
    o = Oracle()
    o.set_video(v.num_frames, v.width, v.height, v.aspect, v.inv_aspect)
    o.set_depth_all(depth)
    o.set_pair_constraints(v.pairs, v.offsets, v.loc, is_static)
    o.reset_depth_xforms(dd)
    o.reset_spatial_xforms(sd)
    o.set_xform_params(dx, False)
    o.set_xform_params(sx, True)
    p = OptParams.defaults()
    p.num_threads = 1
    p.intr_opt = intr
    p.static_loss_type = loss
    extra = {}
    if smooth:
        o.set_triplet_constraints(*trip)
        p.smooth_loss_type, p.smooth_static_weight, p.smooth_dynamic_weight = smooth
        extra = dict(smooth=np.asarray(smooth, dtype=np.float64), trip_centers=trip[0], trip_offsets=trip[1],
                     trip_loc=trip[2], trip_static=trip[3])
    ev = o.evaluate(p, reg, pose, want_gradient=True, want_hdiag=True)
    return dict(**extra, frames=v.num_frames, width=v.width, height=v.height, aspect=np.float32(v.aspect),
                inv_aspect=np.float32(v.inv_aspect), depth=depth, pairs=v.pairs, offsets=v.offsets, loc=v.loc,
                is_static=is_static, pose=pose, depth_params=dx, spatial_params=sx,
                depth_desc=np.frombuffer(bytes(dd), dtype=np.uint8), spatial_desc=np.frombuffer(bytes(sd), dtype=np.uint8),
                intr_opt=intr, loss=loss, depth_deform_reg=reg,
                cost=ev["cost"], num_residual_blocks=ev["num_residual_blocks"], gradient=ev["gradient"], hdiag=ev["hdiag"])


if __name__ == "__main__":
    out = os.path.dirname(os.path.abspath(__file__))
    for name in CASES:
        data = run_oracle(name)
        path = os.path.join(out, name + ".npz")
        np.savez_compressed(path, **data)
        print(f"{name}: cost {data['cost']:.15g}, {data['num_residual_blocks']} residual blocks, "
              f"{os.path.getsize(path) / 1024:.0f} KiB")
