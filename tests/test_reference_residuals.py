"""Reference-held pin of the RESIDUAL ARITHMETIC at random, NON-converged states: the oracle's
three residuals of every static constraint -- ReproDisparity (the default loss), ReproLogDepth and ReproDepthRatio, one case under a
bilinear spatial transform --, and their dual-number Jacobian columns for pose (12), focal lengths (2) and, through d r / d D of the
two deformed depths (2), every depth-transform parameter, against the reference's own torch statement of the same quantities --
utils/geometry.py:62-166 and the reprojection / disparity / depth-ratio terms of loss/consistency_loss.py:93-140 (see
tests/reference_residuals.py for the conversions).  The HIP path's cost and gradient (pose, focal and depth-transform rows) are held
against the same reference outputs with no oracle in between (GPU tests at the end).

CPU only.  Always: against tests/golden/reference_py/residual_golden.npz (outputs of the real reference functions, minted by
make_residual_golden.py next to it).  With /root/reference mounted: against the live functions, including
ConsistencyLoss.geometry_consistency_loss itself.
"""
import os

import numpy as np
import pytest

from robust_cvd_amd.ctypes_types import IntrinsicsOptimization, StaticLossType
from tests import baseline_configs as bc
from tests import reference_residuals as rres

# float64 on both sides: the residual VALUES agree to rounding (measured 5e-14 px on residuals of up to 15 px; 1e-16 on
# disparity differences of up to 0.06); the derivative columns to the accuracy of the central differences (measured 3e-8 on
# entries of up to 330: step 1e-6).  Tolerances sit >= 20 x above the measured worst case.
VALUE_TOL_PX, VALUE_TOL_DISP = 1e-12, 1e-14
FD_REL_TOL = 1e-6


def _golden():
    if not os.path.exists(rres.GOLDEN):
        pytest.skip("residual golden not minted")
    return dict(np.load(rres.GOLDEN))


def _check(name, v, p, pose, sr, ref, rows):
    W, H = v.width, v.height
    r, J = rres.to_reference_units(sr, p, W, H)
    assert np.array_equal(sr["frames"], ref["frames"])
    # the state is not converged: the pin sees residual VALUES, not zeros
    assert np.abs(ref["pixel_diff"]).max() > 5.0 and np.abs(ref["disparity_diff"]).max() > 0.01
    assert np.abs(r[:, :2] - ref["pixel_diff"]).max() < VALUE_TOL_PX
    assert np.abs(r[:, 2] - ref["disparity_diff"]).max() < VALUE_TOL_DISP
    # the reference's loss method: reproj = |pixel difference| / 2, disp = mean(fx, fy over the batch) |disparity difference| / 2
    _ext, intr = rres.cameras(pose, v.aspect, W, H)
    f_mean = intr[sr["frames"][:, 0], :2].mean()
    assert np.abs(np.linalg.norm(r[:, :2], axis=1) / 2.0 - ref["loss_reproj"]).max() < VALUE_TOL_PX
    if rres.CASES[name].get("loss") == StaticLossType.ReproLogDepth:
        # "depth ratio" term: |log(min / max)| / 2 per constraint (lambda = 1, no focal factor: loss/consistency_loss.py:124-140)
        assert np.abs(np.abs(r[:, 2]) / 2.0 - ref["loss_disp"]).max() < 1e-12
        assert (r[:, 2] <= 0.0).all()
    elif rres.CASES[name].get("loss") == StaticLossType.ReproDepthRatio:
        assert (r[:, 2] >= 0.0).all()   # max / min - 1 (no term of this form in the reference's loss method)
    else:
        assert np.abs(f_mean * np.abs(r[:, 2]) / 2.0 - ref["loss_disp"]).max() < 1e-10
    # derivative columns: pose of the source (0-5), pose of the target (6-11), the focal lengths (12, 13)
    fd = ref["fd"]
    scale = np.abs(fd).max(axis=(0, 2), keepdims=True)   # per residual row (pixels / disparity)
    Jr = J[rows]
    assert (np.abs(Jr[:, :, :12] - fd[:, :, :12]) / scale).max() < FD_REL_TOL
    intr_opt = CASE_INTR[name]
    if intr_opt == IntrinsicsOptimization.PerFrame:
        assert (np.abs(Jr[:, :, 12:] - fd[:, :, 12:]) / scale).max() < FD_REL_TOL
    elif intr_opt == IntrinsicsOptimization.Shared:
        # one focal length for every frame (reference lib/PoseOptimizer.cpp:1226): its column is the sum of the two roles'
        both = fd[:, :, 12] + fd[:, :, 13]
        assert (np.abs(Jr[:, :, 12] - both) / scale[:, :, 0]).max() < FD_REL_TOL
        assert np.array_equal(Jr[:, :, 12], Jr[:, :, 13])
    else:
        assert np.abs(Jr[:, :, 12:]).max() == 0.0   # Fixed: no focal column
    # every translation / rotation / focal column carries signal (a column of zeros would pass a relative test vacuously)
    assert (np.abs(fd[:, :2, :12]).max(axis=(0, 1)) > 1.0).all()
    if rres.CASES[name].get("spatial"):
        assert np.abs(sr["cam_a"][:, :2] - sr["ndc_a"]).max() > 0.01   # (the spatial transform really moves the observations)


def _check_depth_columns(name, v, p, pose, sr, sd, fd_depth, rows):
    """The depth-transform columns of the oracle's dual-number Jacobian (round 6): every column of a side is (d r / d D) x the tap's
    factor -- rank one in (residual, tap) to rounding -- and d r / d D of both sides equals the central differences of the
    REFERENCE's functions along the deformed depth they are fed."""
    W, H = v.width, v.height
    ws, wd = p.static_spatial_weight, p.static_depth_weight
    third = 1.0 / wd if p.static_loss_type in (StaticLossType.ReproLogDepth, StaticLossType.ReproDepthRatio) else -1.0 / wd
    unit = np.array([(W / 2.0) / ws, -(H / 2.0) / ws, third])
    J = sd["d_r_d_depth"] * unit[None, :, None]
    assert sd["tap_deviation"].max() < 1e-10, sd["tap_deviation"].max()
    Jr = J[rows]
    scale = np.abs(fd_depth).max(axis=(0, 2), keepdims=True)
    assert (scale > 0).all()
    assert (np.abs(Jr - fd_depth) / scale).max() < FD_REL_TOL, (np.abs(Jr - fd_depth) / scale).max()
    # the source depth moves the re-projected point (all three rows), the target depth only the third residual
    assert np.abs(fd_depth[:, :2, 1]).max() < 1e-6 * scale[:, :2].max() and np.abs(fd_depth[:, 2, 1]).max() > 1e-3
    assert np.abs(fd_depth[:, :2, 0]).max() > 1e-2
    # Scale value transform: D is linear-homogeneous in the side's parameters, so sum_k theta_k d r / d theta_k = D d r / d D
    D = np.stack([sr["cam_a"][:, 2], sr["depth_b"]], 1)
    eul = sd["euler"] * unit[None, :, None]
    assert np.abs(eul - J * D[:, None, :]).max() <= 1e-12 * np.abs(J * D[:, None, :]).max()


CASE_INTR = {k: c["intr"] for k, c in rres.CASES.items()}


@pytest.mark.parametrize("name", sorted(rres.CASES))
def test_oracle_residuals_match_the_committed_reference_outputs(name):
    g = _golden()
    v, p, pose, sr = rres.oracle_side(name)
    assert bc.input_digest(v).encode() == g[name + "/input_sha256"].tobytes(), "the seeded case drifted from the minted one"
    np.testing.assert_array_equal(pose, g[name + "/pose"])
    ref = {k: g[f"{name}/{k}"] for k in ("frames", "pixel_diff", "disparity_diff", "loss_reproj", "loss_disp", "fd")}
    _check(name, v, p, pose, sr, ref, g[name + "/fd_rows"])


@pytest.mark.parametrize("name", sorted(rres.CASES))
def test_oracle_depth_columns_match_the_committed_reference_differences(name):
    from oracle import oracle as om
    g = _golden()
    v, o, p, pose = rres.make_state(name)
    sr = om.static_residuals(o, p, 0.1, pose)
    sd = om.static_residuals_depth(o, p, 0.1, pose)
    _check_depth_columns(name, v, p, pose, sr, sd, g[name + "/fd_depth"], g[name + "/fd_rows"])


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference checkout not mounted")
@pytest.mark.parametrize("name", sorted(rres.CASES))
def test_oracle_residuals_match_the_live_reference_functions(name):
    v, p, pose, sr = rres.oracle_side(name)
    ref = rres.reference_outputs(name)
    _check(name, v, p, pose, sr, ref, np.arange(len(ref["pixel_diff"])))
    from oracle import oracle as om
    _v, o, _p, _pose = rres.make_state(name)
    sd = om.static_residuals_depth(o, p, 0.1, pose)
    _check_depth_columns(name, v, p, pose, sr, sd, ref["fd_depth"], np.arange(len(ref["pixel_diff"])))


def test_the_pin_can_tell_a_wrong_convention():
    """What the comparison would see if the oracle's camera looked down +z, or the image rows ran bottom-up: pixels, not 1e-12."""
    name = "perframe_grid4x3"
    g = _golden()
    v, p, pose, sr = rres.oracle_side(name)
    r, _J = rres.to_reference_units(sr, p, v.width, v.height)
    assert np.abs(-r[:, 1] - g[name + "/pixel_diff"][:, 1]).max() > 1.0     # flipped v
    assert np.abs(-r[:, 2] - g[name + "/disparity_diff"]).max() > 0.01      # flipped disparity sign


def _reference_cost(name, p, v, g):
    logd = rres.loss_kind(name)
    return rres.cost_from_reference_terms(g[name + "/pixel_diff"], g[name + "/disparity_diff"], p, v.width, v.height, logd)


@pytest.mark.parametrize("name", sorted(rres.CASES))
def test_oracle_cost_equals_the_cost_of_the_reference_terms(name):
    """With every regulariser off the problem's cost is 0.5 sum rho_Cauchy(|r|^2) over the static constraints: the oracle's evaluation
    hook against that sum formed from the REFERENCE's per-constraint terms (committed outputs of its torch functions)."""
    g = _golden()
    v, o, p, pose = rres.make_state(name)
    rres.without_regularisers(p)
    ev = o.evaluate(p, 0.0, pose, want_gradient=False)
    assert ev["num_residual_blocks"] == len(g[name + "/pixel_diff"])
    ref = _reference_cost(name, p, v, g)
    assert abs(ev["cost"] - ref) <= 1e-13 * ref, (ev["cost"], ref)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(rres.CASES))
def test_hip_cost_equals_the_cost_of_the_reference_terms(name):
    """The same for the HIP path, with no oracle in between: cvd_evaluate at a random non-converged state against the cost formed
    from what the reference's own geometry.py / consistency_loss.py return for every constraint (fast and generic kernels)."""
    from robust_cvd_amd import api
    g = _golden()
    for generic in (False, True):
        s = api.Solver(0)
        s.set_generic_kernels(generic)
        v, _s, p, pose = rres.make_state(name, s)
        rres.without_regularisers(p)
        ev = s.evaluate(p, 0.0, pose, want_gradient=False)
        ref = _reference_cost(name, p, v, g)
        assert ev["num_residual_blocks"] == len(g[name + "/pixel_diff"])
        assert abs(ev["cost"] - ref) <= 1e-12 * ref, (generic, ev["cost"], ref)
        s.close()


def _check_gradient(name, grad, g):
    """grad [F, B] (layout [t w f | theta | phi]) of the regulariser-free problem against the central differences of the reference cost:
    pose and focal rows, and (round 6) the depth-transform rows."""
    ft = g[name + "/cost_gradient_theta_fd"]
    st = np.abs(ft).max()
    assert st > 0.05   # (the depth rows carry signal as well)
    Gt = grad[:, 7:7 + ft.shape[1]]
    assert np.abs(Gt - ft).max() < 1e-9 * st + 1e-6, (np.abs(Gt - ft).max(), st)
    fd = g[name + "/cost_gradient_fd"]
    scale = np.abs(fd).max()
    assert scale > 10.0   # (the state is far from a minimum: the gradient carries signal)
    G = grad[:, :7].copy()
    intr = rres.CASES[name]["intr"]
    if intr == IntrinsicsOptimization.Shared:
        G[0, 6] = G[:, 6].sum()   # (one focal length: the canonical layout keeps it in frame 0's slot)
        G[1:, 6] = 0.0
    elif intr == IntrinsicsOptimization.Fixed:
        assert np.abs(G[:, 6]).max() == 0.0
    # central differences with step 1e-6 of a cost of O(100): ~1e-8 absolute; measured 5e-8 on entries of up to 785
    assert np.abs(G - fd).max() < 1e-9 * scale + 1e-6, (np.abs(G - fd).max(), scale)


@pytest.mark.parametrize("name", sorted(rres.CASES))
def test_oracle_gradient_equals_the_differences_of_the_reference_cost(name):
    g = _golden()
    v, o, p, pose = rres.make_state(name)
    rres.without_regularisers(p)
    ev = o.evaluate(p, 0.0, pose, want_gradient=True)
    _check_gradient(name, ev["gradient"], g)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(rres.CASES))
def test_hip_gradient_equals_the_differences_of_the_reference_cost(name):
    """J^T (rho' r) of the device's assembly kernels (fast and generic), pose and focal part, against central differences of the
    cost formed from the reference's own per-constraint terms: analytic Jacobians, robust weights and the frame-major accumulation
    of the HIP path held by reference code without the oracle in between."""
    from robust_cvd_amd import api
    g = _golden()
    for generic in (False, True):
        s = api.Solver(0)
        s.set_generic_kernels(generic)
        v, _s, p, pose = rres.make_state(name, s)
        rres.without_regularisers(p)
        ev = s.evaluate(p, 0.0, pose, want_gradient=True)
        _check_gradient(name, ev["gradient"], g)
        s.close()
