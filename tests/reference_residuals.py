"""Shared pieces of the reference-held RESIDUAL pin (Missing #1 / Next #2).

The reference states the ReproDisparity residual of a flow constraint twice: in C++ for Ceres
(lib/PoseOptimizer.cpp:142-319, what the oracle restates and the HIP kernels compute) and in torch for the
fine-tuning loss -- `utils/geometry.py:62-166` (`pixels_to_points`, `reproject_points`, `project`) and the reprojection /
disparity terms of `loss/consistency_loss.py:93-122`.  The torch statement is held by the reference, runs in the build
container and cannot travel to the GPU box, so the pin has two halves like the output-convention pin next to it:

  * tests/golden/reference_py/make_residual_golden.py runs the REAL torch functions (float64) on seeded random,
    NON-converged states and commits what they return (residual_golden.npz): per constraint the pixel difference, the
    disparity difference, the two terms of `ConsistencyLoss.geometry_consistency_loss` itself, and central differences of
    the torch functions along every pose / focal parameter;
  * tests/test_reference_residuals.py holds the oracle's three residuals and its dual-number Jacobian columns against the
    committed values (always) and against the live reference functions (when /root/reference is mounted).

Conversions (derived from the two statements, checked by the test itself):
    pixel = ((ndc.x + 1) W/2, (1 - ndc.y) H/2)                      (pixel-edge NDC of the constraints)
    intrinsics = (W/2 / (vfocal aspect), H/2 / vfocal, W/2, H/2)     (update_poses, loaders/video_dataset.py:185-190)
    extrinsics = [R(angle-axis) | t], columns right / up / backward  (update_poses :179-182; R by scipy's Rodrigues)
    r_x / ws =  (pixels_tgt - matched).x / (W/2),   r_y / ws = -(pixels_tgt - matched).y / (H/2)
    r_z / wd = -(1 / z_tgt - 1 / z_warped)          (the reference's camera looks down -z: z = -depth)

What the pin covers: a6 / a7 of SURVEY.md 8 -- the camera model, pose and projection conventions, the three depth terms
(disparity, log depth, depth ratio) -- as VALUES and first derivatives away from the optimum: along pose (12), focal lengths (2)
and, since round 6, along the two deformed depths D_a, D_b (2), which is what every depth-transform column of the optimizer's
Jacobian is made of (d r / d theta_k = (d r / d D) w_k d_src); one case runs under a non-trivial bilinear spatial transform.  What it
takes as given: the gather weights w_k of the depth / spatial functors (pinned by the goldens of tests/golden/; the reference samples
a network output there).
"""
import os
import sys
import types

import numpy as np

from robust_cvd_amd import synth
from robust_cvd_amd.ctypes_types import IntrinsicsOptimization, OptParams, StaticLossType, XformDesc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_py", "residual_golden.npz")

# name -> (frames, width, height, seed, intrinsics mode, depth grid)
CASES = {
    "perframe_grid4x3": dict(frames=8, width=96, height=56, seed=515, intr=IntrinsicsOptimization.PerFrame, grid=(4, 3)),
    "fixed_global": dict(frames=6, width=64, height=48, seed=516, intr=IntrinsicsOptimization.Fixed, grid=None),
    "shared_grid3x3_portrait": dict(frames=6, width=48, height=80, seed=517, intr=IntrinsicsOptimization.Shared, grid=(3, 3)),
    # ReproLogDepth: third residual = log(min / max) of the reprojected and the target's depth -- the reference's "depth ratio"
    # term (loss/consistency_loss.py:124-140), the second of the optimizer's four static losses the reference's Python states
    "perframe_logdepth": dict(frames=6, width=64, height=48, seed=518, intr=IntrinsicsOptimization.PerFrame, grid=(4, 3),
                              loss=StaticLossType.ReproLogDepth),
    # ReproDepthRatio (round 6): third residual = max / min - 1 of the same two depths (reference lib/PoseOptimizer.cpp:294-297); the
    # Python reference has no term of this form, but the two depths it is made of are outputs of its functions
    "perframe_ratio": dict(frames=6, width=64, height=48, seed=519, intr=IntrinsicsOptimization.PerFrame, grid=(4, 3),
                           loss=StaticLossType.ReproDepthRatio),
    # a bilinear SPATIAL transform with random non-zero parameters (round 6): both observations' warped NDC enter the camera model
    # (reference lib/PoseOptimizer.cpp:162-175, lib/DepthMapTransform.cpp:1253-1343), i.e. the reference's functions see shifted pixels
    "perframe_spatial3x2": dict(frames=6, width=64, height=48, seed=520, intr=IntrinsicsOptimization.PerFrame, grid=(4, 3),
                                spatial=(3, 2)),
}
FD_STEP = 1e-6   # central differences of the torch functions (float64)


def make_state(name, binding=None):
    """A seeded NON-converged state of a small synthetic problem: the true cameras perturbed by centimetres / degrees /
    10 % in focal length, flow noise 1 px, random depth scales -- residuals of O(0.01 .. 1) NDC units.  `binding`: the object
    to load it into (default: a fresh Oracle; the GPU test passes the product Solver)."""
    c = CASES[name]
    v = synth.make_video(c["frames"], c["width"], c["height"], seed=c["seed"], flow_noise_px=1.0, spacing=9.0)
    if binding is None:
        from oracle.oracle import Oracle
        binding = Oracle()
    o = binding
    synth.load_into(o, v)
    o.reset_depth_xforms(XformDesc.grid_depth(*c["grid"]) if c["grid"] else XformDesc.global_depth())
    if c.get("spatial"):
        from robust_cvd_amd.ctypes_types import SpatialXformType
        o.reset_spatial_xforms(XformDesc.spatial(SpatialXformType.BilinearGrid, *c["spatial"]))
    else:
        o.reset_spatial_xforms(XformDesc.spatial())
    rng = np.random.default_rng(c["seed"] + 7)
    F = v.num_frames
    pose = np.zeros((F, 7))
    pose[:, :3] = v.true_t + rng.normal(0, 0.03, (F, 3))
    pose[:, 3:6] = v.true_w + rng.normal(0, np.deg2rad(1.5), (F, 3))
    pose[:, 6] = v.true_fy * (1.0 + rng.uniform(-0.1, 0.1, F))
    if c["intr"] == IntrinsicsOptimization.Shared:
        pose[:, 6] = pose[0, 6]
    dx = o.get_xform_params(False)
    dx = v.frame_scale[:, None] * (1.0 + rng.uniform(-0.15, 0.15, dx.shape))
    o.set_xform_params(dx, False)
    if c.get("spatial"):
        sx = o.get_xform_params(True)
        o.set_xform_params(rng.normal(0.0, 0.02, sx.shape), True)    # NDC units: ~1 px at 64 x 48
    p = OptParams.defaults()
    p.num_threads = 1
    p.intr_opt = int(c["intr"])
    p.static_spatial_weight = 1.3   # (not 1: the weights must not hide in the comparison)
    p.static_depth_weight = 0.7
    p.static_loss_type = int(c.get("loss", StaticLossType.ReproDisparity))
    if c["intr"] == IntrinsicsOptimization.Fixed:
        pose[:, 6] = v.true_fy      # (what the optimizer uses there: vFocal(params), reference lib/PoseOptimizer.cpp:1155-1157)
    return v, o, p, pose


def oracle_side(name):
    from oracle import oracle as om
    v, o, p, pose = make_state(name)
    sr = om.static_residuals(o, p, 0.1, pose)
    return v, p, pose, sr


def cameras(pose, aspect, W, H):
    """update_poses' tensors from the optimizer's 7-tuples (float64): extrinsics [F, 3, 4], intrinsics [F, 4]."""
    from scipy.spatial.transform import Rotation
    F = pose.shape[0]
    ext = np.zeros((F, 3, 4))
    ext[:, :, :3] = Rotation.from_rotvec(pose[:, 3:6]).as_matrix()   # columns: right, up, backward
    ext[:, :, 3] = pose[:, :3]
    intr = np.zeros((F, 4))
    intr[:, 0] = (W / 2.0) / (pose[:, 6] * aspect)    # W/2 / tan(hFov/2), tan(hFov/2) = vfocal * aspect
    intr[:, 1] = (H / 2.0) / pose[:, 6]
    intr[:, 2], intr[:, 3] = W / 2.0, H / 2.0
    return ext, intr


def to_pixels(ndc, W, H):
    return np.stack([(ndc[:, 0] + 1.0) * (W / 2.0), (1.0 - ndc[:, 1]) * (H / 2.0)], 1)


def _reference_modules():
    cv2 = types.ModuleType("cv2")
    cv2.CV_32FC3, cv2.CV_8UC1, cv2.IMREAD_UNCHANGED = 21, 0, -1
    sys.modules.setdefault("cv2", cv2)
    # (loaders/video_dataset.py imports the enum classes `from lib_python`: the drop-in module's own)
    from robust_cvd_amd import build as b
    d = os.path.dirname(b.lib_python_path())
    if d not in sys.path:
        sys.path.insert(0, d)
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    from utils import geometry
    from loss.consistency_loss import ConsistencyLoss
    return geometry, ConsistencyLoss


def loss_kind(name):
    """False: ReproDisparity; True: ReproLogDepth; "ratio": ReproDepthRatio (the `log_depth` argument of the functions below)."""
    loss = CASES[name].get("loss")
    return "ratio" if loss == StaticLossType.ReproDepthRatio else (loss == StaticLossType.ReproLogDepth)


def reference_terms(geometry, ext, intr, fa, fb, pix_a, depth_a, pix_b, depth_b, log_depth=False, ext_b=None, intr_b=None):
    """The reference's functions, one constraint per batch entry as (n, C, 1, 1) float64 tensors: returns the pixel difference
    `project(reproject_points(pixels_to_points(..)))  - (pixels + flow)` [n, 2] and the disparity difference
    `1 / z_tgt - 1 / z_warped` [n] (consistency_loss.py:99-103, :117-118; `sample` of the target's point map at the matched pixel
    is the target observation's own camera-space point)."""
    import torch
    n = len(fa)
    ext_b = ext if ext_b is None else ext_b      # (the target's camera tensors from another state: one-sided differences)
    intr_b = intr if intr_b is None else intr_b
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64)
    points_ref = geometry.pixels_to_points(t(intr[fa]), t(depth_a).view(n, 1, 1, 1), t(pix_a).view(n, 2, 1, 1).clone())
    points_tgt = geometry.reproject_points(points_ref, t(ext[fa]), t(ext_b[fb]))
    pixels_tgt = geometry.project(points_tgt, t(intr_b[fb]))
    matched = t(pix_b).view(n, 2, 1, 1)
    warped_tgt = geometry.pixels_to_points(t(intr_b[fb]), t(depth_b).view(n, 1, 1, 1), matched.clone())
    if isinstance(log_depth, str) and log_depth == "ratio":
        # ReproDepthRatio, lib/PoseOptimizer.cpp:294-297: max / min - 1 of the same two depths (formed here from the functions' outputs)
        dw, dt = torch.abs(warped_tgt[:, -1:, ...]), torch.abs(points_tgt[:, -1:, ...])
        third = torch.max(dw, dt) / torch.min(dw, dt) - 1.0
    elif log_depth:
        # loss/consistency_loss.py:130-137: log(min / max) of |z| of the warped target point and of the reprojected point
        dw, dt = torch.abs(warped_tgt[:, -1:, ...]), torch.abs(points_tgt[:, -1:, ...])
        third = torch.log(torch.min(dw, dt) / torch.max(dw, dt))
    else:
        third = 1.0 / points_tgt[:, -1:, ...] - 1.0 / warped_tgt[:, -1:, ...]
    return (pixels_tgt - matched).view(n, 2).numpy(), third.view(n).numpy()


def reference_loss_terms(ConsistencyLoss, ext, intr, fa, fb, pix_a, depth_a, pix_b, depth_b, log_depth=False):
    """`ConsistencyLoss.geometry_consistency_loss` ITSELF (loss/consistency_loss.py:27-199) with the l1 distance, one
    constraint per batch entry as a constant 2 x 2 image pair; the reverse direction is masked out, so the method returns
    per constraint  reproj = |pixel difference| / 2  and  disp = mean(fx, fy) |disparity difference| / 2."""
    import torch
    import utils.torch_helpers as th
    n = len(fa)
    opt = types.SimpleNamespace(distance_type_static="l1", distance_scale=1.0, lambda_static_reprojection=1.0,
                                lambda_static_disparity=0.0 if log_depth else 1.0, lambda_static_depth_ratio=1.0 if log_depth else 0.0)
    loss = ConsistencyLoss(opt)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64)
    img = lambda a, c: t(a).view(n, c, 1, 1).expand(n, c, 2, 2).contiguous()
    from utils import geometry
    pix = torch.stack([img(pix_a, 2), img(pix_b, 2)], 1)                                     # (n, 2, 2, H, W)
    pts = torch.stack([geometry.pixels_to_points(t(intr[fa]), img(depth_a, 1), img(pix_a, 2).clone()),
                       geometry.pixels_to_points(t(intr[fb]), img(depth_b, 1), img(pix_b, 2).clone())], 1)
    meta = {"extrinsics": torch.stack([t(ext[fa]), t(ext[fb])], 1), "intrinsics": torch.stack([t(intr[fa]), t(intr[fb])], 1),
            "geometry_consistency": {"flows": (img(pix_b - pix_a, 2), img(pix_a - pix_b, 2)),
                                     "masks": (torch.ones(n, 1, 2, 2, dtype=torch.float64),
                                               torch.zeros(n, 1, 2, 2, dtype=torch.float64))}}
    assert th._device.type == "cpu"
    _total, batch = loss.geometry_consistency_loss(pts, meta, pix)
    return batch["reproj"].numpy(), batch["depth ratio" if log_depth else "disp"].numpy()


def reference_outputs(name):
    """Everything the fixture holds for one case (needs /root/reference)."""
    geometry, ConsistencyLoss = _reference_modules()
    v, p, pose, sr = oracle_side(name)
    W, H = v.width, v.height
    fa, fb = sr["frames"][:, 0], sr["frames"][:, 1]
    pix_a = to_pixels(sr["cam_a"][:, :2], W, H)      # the WARPED NDC of both observations (identity spatial transform: the NDC)
    pix_b = to_pixels(sr["cam_b"][:, :2], W, H)
    Da, Db = sr["cam_a"][:, 2], sr["depth_b"]

    log_depth = loss_kind(name)
    ext, intr = cameras(pose, v.aspect, W, H)
    dpx, ddisp = reference_terms(geometry, ext, intr, fa, fb, pix_a, Da, pix_b, Db, log_depth)
    if log_depth == "ratio":   # (no term of this form in the Python reference: the loss method is not called)
        l_reproj, l_disp = np.linalg.norm(dpx, axis=1) / 2.0, np.zeros(len(fa))
    else:
        l_reproj, l_disp = reference_loss_terms(ConsistencyLoss, ext, intr, fa, fb, pix_a, Da, pix_b, Db, log_depth)
    # central differences of the torch functions along the 7 parameters of frame a resp. frame b, per constraint:
    # perturbing parameter k of EVERY frame that is a constraint's source (resp. target) at once is not the same thing when a
    # frame is both, so source and target are perturbed through two copies of the camera tensors
    n = len(fa)
    fd = np.zeros((n, 3, 14))
    cols = [(0, k, k) for k in range(6)] + [(1, k, 6 + k) for k in range(6)] + [(0, 6, 12), (1, 6, 13)]
    for which, k, col in cols:
        out = []
        for sgn in (+1.0, -1.0):
            ps = pose.copy()
            ps[:, k] += sgn * FD_STEP
            e1, i1 = cameras(ps, v.aspect, W, H)
            ea, ia = (e1, i1) if which == 0 else (ext, intr)
            eb, ib = (e1, i1) if which == 1 else (ext, intr)
            out.append(reference_terms(geometry, ea, ia, fa, fb, pix_a, Da, pix_b, Db, log_depth, ext_b=eb, intr_b=ib))
        fd[:, 0, col] = (out[0][0][:, 0] - out[1][0][:, 0]) / (2 * FD_STEP)
        fd[:, 1, col] = (out[0][0][:, 1] - out[1][0][:, 1]) / (2 * FD_STEP)
        fd[:, 2, col] = (out[0][1] - out[1][1]) / (2 * FD_STEP)
    # ... and along the two deformed depths (round 6: what the theta columns of the optimizer's Jacobian are made of,
    # d r / d theta_k = (d r / d D) w_k d_src): relative steps, one constraint per batch entry
    fd_depth = np.zeros((n, 3, 2))
    for side in range(2):
        out = []
        for sgn in (+1.0, -1.0):
            da = Da * (1.0 + sgn * FD_STEP) if side == 0 else Da
            db = Db * (1.0 + sgn * FD_STEP) if side == 1 else Db
            out.append(reference_terms(geometry, ext, intr, fa, fb, pix_a, da, pix_b, db, log_depth))
        h = 2 * FD_STEP * (Da if side == 0 else Db)
        fd_depth[:, 0, side] = (out[0][0][:, 0] - out[1][0][:, 0]) / h
        fd_depth[:, 1, side] = (out[0][0][:, 1] - out[1][0][:, 1]) / h
        fd_depth[:, 2, side] = (out[0][1] - out[1][1]) / h
    return dict(pixel_diff=dpx, disparity_diff=ddisp, loss_reproj=l_reproj, loss_disp=l_disp, fd=fd, fd_depth=fd_depth,
                frames=sr["frames"], pose=pose)


def to_reference_units(sr, p, W, H):
    """The oracle's residuals / Jacobian rows in the reference's units (pixels, disparity): see the module docstring.
    (ReproLogDepth: the third residual IS log(min / max) times the depth weight -- no sign change.)"""
    ws, wd = p.static_spatial_weight, p.static_depth_weight
    third = 1.0 / wd if p.static_loss_type in (StaticLossType.ReproLogDepth, StaticLossType.ReproDepthRatio) else -1.0 / wd
    s = np.array([(W / 2.0) / ws, -(H / 2.0) / ws, third])
    return sr["residuals"] * s[None, :], sr["jacobian"] * s[None, :, None]


def cost_from_reference_terms(pixel_diff, third, p, W, H, log_depth=False):
    """The static part of the optimizer's cost, 0.5 sum rho_Cauchy(|r|^2) (ceres::CauchyLoss(robustness): rho(s) = b^2 log(1 + s / b^2),
    reference lib/PoseOptimizer.cpp:1220), from the REFERENCE's per-constraint pixel / disparity (or log-depth) terms."""
    ws, wd, b2 = p.static_spatial_weight, p.static_depth_weight, p.robustness ** 2
    r0 = pixel_diff[:, 0] * ws / (W / 2.0)
    r1 = -pixel_diff[:, 1] * ws / (H / 2.0)
    r2 = third * wd * (1.0 if log_depth else -1.0)   # (log depth and ratio enter as they are; the disparity term changes sign)
    s = r0 * r0 + r1 * r1 + r2 * r2
    return 0.5 * float(np.sum(b2 * np.log1p(s / b2)))


def without_regularisers(p):
    """The same problem with every regulariser switched off: the cost is then the static constraints' alone."""
    p.scale_reg = 0.0
    p.focal_reg = 0.0
    p.depth_deform_reg_initial = p.depth_deform_reg_final = 0.0
    p.spatial_deform_reg = 0.0
    p.position_reg = 0.0
    return p


def reference_cost_gradient(name):
    """Central differences (step FD_STEP) of the REFERENCE cost -- 0.5 sum rho_Cauchy(|r|^2) with r from the reference's torch
    functions -- along every frame's pose parameters [t(3) w(3)] and focal length: [F, 7].  Shared intrinsics: the one focal length
    is perturbed in every frame at once and its derivative is stored in frame 0's slot; Fixed: the focal column stays 0.
    (needs /root/reference)"""
    geometry, _cl = _reference_modules()
    v, p, pose, sr = oracle_side(name)
    without_regularisers(p)
    W, H = v.width, v.height
    fa, fb = sr["frames"][:, 0], sr["frames"][:, 1]
    pix_a, pix_b = to_pixels(sr["cam_a"][:, :2], W, H), to_pixels(sr["cam_b"][:, :2], W, H)
    Da, Db = sr["cam_a"][:, 2], sr["depth_b"]
    log_depth = loss_kind(name)

    def cost(ps):
        ext, intr = cameras(ps, v.aspect, W, H)
        d, t = reference_terms(geometry, ext, intr, fa, fb, pix_a, Da, pix_b, Db, log_depth)
        return cost_from_reference_terms(d, t, p, W, H, log_depth)

    F = v.num_frames
    g = np.zeros((F, 7))
    intr_mode = CASES[name]["intr"]
    for f in range(F):
        for k in range(7 if intr_mode == IntrinsicsOptimization.PerFrame else 6):
            a, b = pose.copy(), pose.copy()
            a[f, k] += FD_STEP
            b[f, k] -= FD_STEP
            g[f, k] = (cost(a) - cost(b)) / (2 * FD_STEP)
    if intr_mode == IntrinsicsOptimization.Shared:
        a, b = pose.copy(), pose.copy()
        a[:, 6] += FD_STEP
        b[:, 6] -= FD_STEP
        g[0, 6] = (cost(a) - cost(b)) / (2 * FD_STEP)
    return g


def reference_cost_gradient_theta(name):
    """Central differences of the REFERENCE cost along every depth-transform parameter: [F, nD].  The deformed depths D_a, D_b that the
    reference's functions are fed come from the oracle's depth functor at the perturbed parameters (the gathers are pinned by the goldens
    of tests/golden/); everything downstream of them -- back-projection, re-projection, the three residual terms, the robust loss --
    is the reference's own arithmetic.  (needs /root/reference)"""
    from oracle import oracle as om
    geometry, _cl = _reference_modules()
    v, o, p, pose = make_state(name)
    without_regularisers(p)
    W, H = v.width, v.height
    log_depth = loss_kind(name)
    ext, intr = cameras(pose, v.aspect, W, H)
    theta0 = o.get_xform_params(False).copy()

    def cost(theta):
        o.set_xform_params(theta, False)
        sr = om.static_residuals(o, p, 0.0, pose)
        fa, fb = sr["frames"][:, 0], sr["frames"][:, 1]
        pix_a, pix_b = to_pixels(sr["cam_a"][:, :2], W, H), to_pixels(sr["cam_b"][:, :2], W, H)
        d, t = reference_terms(geometry, ext, intr, fa, fb, pix_a, sr["cam_a"][:, 2], pix_b, sr["depth_b"], log_depth)
        return cost_from_reference_terms(d, t, p, W, H, log_depth)

    g = np.zeros_like(theta0)
    for f in range(theta0.shape[0]):
        for k in range(theta0.shape[1]):
            a, b = theta0.copy(), theta0.copy()
            h = FD_STEP * max(1.0, abs(theta0[f, k]))
            a[f, k] += h
            b[f, k] -= h
            g[f, k] = (cost(a) - cost(b)) / (2 * h)
    o.set_xform_params(theta0, False)
    return g
