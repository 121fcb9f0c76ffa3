"""Shared pieces of the reference-held OUTPUT-CONVENTION pin (Missing #3).

The reference consumes the optimizer's results in Python: `VideoDataset.update_poses` (reference
loaders/video_dataset.py:153-217) reads `extrinsics.right()/up()/backward()/position`, `intrinsics.hFov/vFov`,
`depthXform().paramMap(frame)` and `spatialXform().warp(h, w)` from `lib_python` -- the drop-in -- and
`utils/geometry.py:86-166` (`pixels_to_points`, `points_cam_to_world`, `world_to_points_cam`, `project`) re-derives the
camera model from them.  That code is held by the reference, runs in the build container (torch only; cv2 stubbed) and
cannot travel to the GPU box, so the pin has two halves:

  * tests/golden/reference_py/make_reprojection_golden.py runs the drop-in on the GPU box (`dump`), then -- in the
    container -- replays the dumped getter values through the REAL reference functions (`mint`) and commits what they
    return (extrinsics / intrinsics tensors, scale maps, reprojected pixels of every static constraint);
  * the functions below restate those few reference lines in numpy.  tests/test_reference_reprojection.py holds the
    restatement against the committed reference outputs (CPU) and applies it to a fresh drop-in run (GPU).

Nothing here is part of the optimizer arithmetic.
"""
import os

import numpy as np

from robust_cvd_amd import dataset_io, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_py", "reprojection_golden.npz")

CASE = dict(frames=16, width=192, height=112, seed=4242, ctf=(4, 3), field_amp=0.10, deform_reg=1e-4, scale_reg=1e-6)


def grid_field(theta, W, H):
    """theta [gy, gx] (row 0 = image BOTTOM) evaluated per pixel with the optimizer's own bilinear gather (reference
    lib/DepthMapTransform.cpp:750-764, 823-840: s = (n + 1)(g - 1)/2 at the pixel-edge NDC n.x = -1 + 2 i / W,
    n.y = 1 - 2 j / H of the constraints)."""
    gy, gx = theta.shape
    nx = -1.0 + 2.0 * np.arange(W) / W
    ny = 1.0 - 2.0 * np.arange(H) / H
    sx = np.clip((nx + 1.0) * (gx - 1) / 2.0, 0.0, np.nextafter(gx - 1.0, 0.0))
    sy = np.clip((ny + 1.0) * (gy - 1) / 2.0, 0.0, np.nextafter(gy - 1.0, 0.0))
    ix, iy = sx.astype(int), sy.astype(int)
    rx, ry = (sx - ix)[None, :], (sy - iy)[:, None]
    a, b = theta[iy][:, ix], theta[iy][:, ix + 1]
    c, d = theta[iy + 1][:, ix], theta[iy + 1][:, ix + 1]
    return (1 - ry) * ((1 - rx) * a + rx * b) + ry * ((1 - rx) * c + rx * d)


def make_case():
    """Zero-noise video whose per-frame depth error is EXACTLY what the final level of the case's schedule can undo: the depth
    handed to the optimizer is the rendered depth divided by s_f * theta_f(x, y), theta_f a random 4 x 3 bilinear grid in the
    optimizer's own gather convention.  An exact solution exists (the deformation regulariser, which pulls the vertices
    together, and the scale regulariser, which pulls every vertex towards 1 / median depth, are set to 1e-4 / 1e-6 for this
    case: at their defaults they outweigh the noise-free data term and the optimum is visibly not the true field), the end state reprojects every static constraint onto its flow target, and the scale
    MAP varies by +-10 % over the image -- top to bottom too, so the row order of paramMap matters to the result."""
    c = CASE
    v = synth.make_video(c["frames"], c["width"], c["height"], seed=c["seed"], flow_noise_px=0.0, field_amp=0.0)
    rng = np.random.default_rng(c["seed"] + 1)
    gx, gy = c["ctf"]
    v.true_theta = 1.0 + c["field_amp"] * rng.uniform(-1.0, 1.0, size=(v.num_frames, gy, gx))
    for f in range(v.num_frames):
        v.depth[f] = (v.depth[f].astype(np.float64) / grid_field(v.true_theta[f], v.width, v.height)).astype(np.float32)
    return v


def run_drop_in(lib, video, base_dir):
    """pose_optimization.py's sequence on the drop-in module (tests/drop_in_caller.py), then every getter update_poses reads."""
    from tests.drop_in_caller import build_pose_optimizer, optimize_poses
    base = dataset_io.write_dataset(base_dir, video)
    frames = list(range(video.num_frames))
    opt = lib.DepthVideoPoseOptimizer.Params()
    opt.ctfLong, opt.ctfShort = CASE["ctf"]
    opt.depthDeformRegInitial = opt.depthDeformRegFinal = CASE["deform_reg"]
    opt.scaleReg = CASE["scale_reg"]
    dv, fc = build_pose_optimizer(lib, base, "midas2", frames, opt)
    optimize_poses(lib, dv, fc, frames, opt)
    ds = dv.depthStream(dv.numDepthStreams() - 1)
    out = {k: [] for k in ("right", "up", "backward", "position", "orientation", "hfov", "vfov", "param_map", "params", "warp",
                           "source_depth")}
    for i in frames:
        f = ds.frame(i)
        out["right"].append(np.asarray(f.extrinsics.right(), np.float64))
        out["up"].append(np.asarray(f.extrinsics.up(), np.float64))
        out["backward"].append(np.asarray(f.extrinsics.backward(), np.float64))
        out["position"].append(np.asarray(f.extrinsics.position, np.float64))
        out["orientation"].append(np.asarray(f.extrinsics.orientation.coeffs(), np.float64))
        out["hfov"].append(float(f.intrinsics.hFov))
        out["vfov"].append(float(f.intrinsics.vFov))
        out["param_map"].append(np.asarray(f.depthXform().paramMap(f), np.float64))
        out["params"].append(np.asarray(f.depthXform().params(), np.float64))
        out["warp"].append(np.asarray(f.spatialXform().warp(ds.height(), ds.width()), np.float32))
        out["source_depth"].append(np.asarray(f.sourceDepth(), np.float32))
    out = {k: np.stack(v) for k, v in out.items()}
    out["width"], out["height"] = np.int32(ds.width()), np.int32(ds.height())
    out["depth_desc"] = np.frombuffer(ds.depthXformDesc().str().encode(), np.uint8)
    return out


def numpy_update_poses(out):
    """VideoDataset.update_poses (reference loaders/video_dataset.py:170-215) on the dumped getter values:
    extrinsics [N, 3, 4] = [right | up | backward | position], intrinsics [N, 4] = (W/2 / tan(hFov/2), H/2 / tan(vFov/2),
    W/2, H/2), float32 like the reference's `_dtype`."""
    N = out["position"].shape[0]
    W, H = float(out["width"]) / 2.0, float(out["height"]) / 2.0
    ext = np.zeros((N, 3, 4), np.float32)
    ext[:, :, 0], ext[:, :, 1], ext[:, :, 2], ext[:, :, 3] = out["right"], out["up"], out["backward"], out["position"]
    intr = np.zeros((N, 4), np.float32)
    intr[:, 0] = W / np.tan(np.asarray(out["hfov"], np.float64) / 2.0)
    intr[:, 1] = H / np.tan(np.asarray(out["vfov"], np.float64) / 2.0)
    intr[:, 2], intr[:, 3] = W, H
    return ext, intr


def constraint_samples(video, out):
    """Per static constraint: frames (a, b), the source pixel (integer, x right / y down, top-left origin), the flow target
    in pixels and the source depth at the truncating fetch of reference lib/PoseOptimizer.cpp:113-115."""
    W, H = video.width, video.height
    fa = np.repeat(video.pairs[:, 0], np.diff(video.offsets))
    fb = np.repeat(video.pairs[:, 1], np.diff(video.offsets))
    sy = np.float32(H) / np.float32(video.inv_aspect)
    pa = np.stack([video.loc[:, 0] * np.float32(W), video.loc[:, 1] * sy], 1).astype(np.float64)
    pb = np.stack([video.loc[:, 2] * np.float32(W), video.loc[:, 3] * sy], 1).astype(np.float64)
    ix = np.clip(np.rint(pa[:, 0]).astype(int), 0, W - 1)  # (loc0 is an integer pixel times 1/W: rint undoes the f32 rounding)
    iy = np.clip(np.rint(pa[:, 1]).astype(int), 0, H - 1)
    depth = out["source_depth"][fa, iy, ix].astype(np.float64) * out["param_map"][fa, iy, ix]
    return fa, fb, np.stack([ix, iy], 1).astype(np.float64), pb, depth


def numpy_reproject(ext, intr, fa, fb, pix, depth):
    """utils/geometry.py: pixels_to_points (:86-100: rays ((x - cx)/fx, -(y - cy)/fy, -1) * depth), points_cam_to_world
    (:103-123: t + R p), world_to_points_cam (:126-138: R^T (p - t)) and project (:62-83: (u, -v) * f + c), float32 like torch."""
    ext = ext.astype(np.float32)
    intr = intr.astype(np.float32)
    pix = pix.astype(np.float32)
    uv = pix - intr[fa, 2:]
    uv[:, 1] = -uv[:, 1]
    rays = np.concatenate([uv / intr[fa, :2], -np.ones((len(fa), 1), np.float32)], 1)
    pts = rays * depth.astype(np.float32)[:, None]
    world = ext[fa, :, 3] + np.einsum("nij,nj->ni", ext[fa, :, :3], pts)
    cam = np.einsum("nji,nj->ni", ext[fb, :, :3], world - ext[fb, :, 3])
    r = cam / -cam[:, 2:3]
    uv2 = r[:, :2] * intr[fb, :2]
    uv2[:, 1] = -uv2[:, 1]
    return uv2 + intr[fb, 2:]


def depth_scale_spread(video, out):
    """depth x paramMap against the rendered depth: the per-frame medians of the ratio must agree up to ONE global scale
    (the gauge); returns the largest relative deviation of a frame's median ratio from the global median."""
    ratio = out["source_depth"].astype(np.float64) * out["param_map"] / video.true_depth.astype(np.float64)
    per_frame = np.median(ratio.reshape(ratio.shape[0], -1), axis=1)
    g = np.median(per_frame)
    return float(np.abs(per_frame / g - 1.0).max()), float(np.abs(ratio / g - 1.0).max())
