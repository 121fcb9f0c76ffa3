"""GPU parity (-m gpu) of the Huber stress variant (cvd_solver_options::robust_loss = 1, BASELINE.json configs[4]):
ceres::HuberLoss(robustness) on the static flow constraints instead of the reference's CauchyLoss.

Same bars as tests/test_gpu_parity.py: cost / gradient / J^T J to 1e-9 relative against the oracle (f64 on both sides),
converged solves to 1e-4 relative final cost and gauge-aligned poses.
"""
import numpy as np
import pytest

from oracle.oracle import Oracle
from robust_cvd_amd import synth
from robust_cvd_amd.ctypes_types import OptParams, SpatialXformType, StaticLossType, ValueXformType, XformDesc
from tests.helpers import rel

pytestmark = pytest.mark.gpu

TOL = 1e-9

CASES = [
    # (depth transform, spatial transform, static loss, robustness): fast kernels (Global / bilinear / bicubic grids with an
    # identity spatial transform) and the generic all-variants kernels
    ("global", XformDesc.global_depth(), XformDesc.spatial(), StaticLossType.ReproDisparity, 1.0),
    ("grid5x4", XformDesc.grid_depth(5, 4), XformDesc.spatial(), StaticLossType.ReproDisparity, 0.6),
    ("cubic4x4", XformDesc.grid_depth(4, 4, cubic=True), XformDesc.spatial(), StaticLossType.ReproDepthRatio, 1.0),
    ("grid17x10", XformDesc.grid_depth(17, 10), XformDesc.spatial(), StaticLossType.ReproLogDepth, 1.0),
    ("generic_ss_spatial", XformDesc.grid_depth(4, 3, ValueXformType.ScaleShift, cubic=True),
     XformDesc.spatial(SpatialXformType.BilinearGrid, 3, 2), StaticLossType.ReproDisparity, 1.0),
    ("generic_euclid", XformDesc.global_depth(), XformDesc.spatial(), StaticLossType.Euclidean, 0.4),
]


@pytest.fixture(scope="module")
def Solver():
    from robust_cvd_amd import api
    return api.Solver


def _state(s, F, rng):
    pose = np.zeros((F, 7))
    pose[:, :6] = rng.normal(0, 0.05, (F, 6))
    pose[:, 6] = 0.2 + rng.uniform(0, 0.02, F)
    dx = s.get_xform_params()
    if dx.size:
        dx = 0.15 + rng.uniform(0, 0.05, dx.shape)
        if s.xform_desc().value_xform == 2:
            dx[:, 1::2] = rng.uniform(0, 0.3, dx[:, 1::2].shape)
    sx = rng.normal(0, 0.01, s.get_xform_params(True).shape)
    return pose, dx, sx


@pytest.mark.parametrize("name,ddesc,sdesc,loss,robustness", CASES, ids=[c[0] for c in CASES])
def test_huber_cost_gradient_hessian_match_oracle(Solver, name, ddesc, sdesc, loss, robustness):
    v = synth.make_video(4, 64, 40, seed=41, spacing=9)
    p = OptParams.defaults()
    p.num_threads = 2
    p.static_loss_type = loss
    p.robustness = robustness
    res, state = {}, None
    for k, ctor in (("oracle", Oracle), ("hip", lambda: Solver(0)), ("hip_cauchy", lambda: Solver(0))):
        s = ctor()
        synth.load_into(s, v)
        s.reset_depth_xforms(ddesc)
        s.reset_spatial_xforms(sdesc)
        if state is None:
            state = _state(s, v.num_frames, np.random.default_rng(17))
        pose, dx, sx = state
        if dx.size:
            s.set_xform_params(dx)
        if sx.size:
            s.set_xform_params(sx, True)
        s.set_robust_loss(0 if k == "hip_cauchy" else 1)
        res[k] = s.evaluate(p, 0.1, pose, want_hdiag=True, want_hfull=True)
    h, o = res["hip"], res["oracle"]
    assert h["num_residual_blocks"] == o["num_residual_blocks"]
    assert abs(h["cost"] - o["cost"]) <= TOL * abs(o["cost"])
    assert rel(h["gradient"], o["gradient"]) < TOL
    assert rel(h["hdiag"], o["hdiag"]) < TOL
    assert rel(h["hfull"], o["hfull"]) < TOL          # matrix-free product, column by column
    # the option does something: the Cauchy cost at the same point differs (both loss branches are populated)
    assert abs(res["hip_cauchy"]["cost"] - h["cost"]) > 1e-3 * abs(h["cost"])


def test_huber_option_is_validated(Solver):
    s = Solver(0)
    v = synth.make_video(3, 48, 28, seed=4, spacing=8)
    synth.load_into(s, v)
    s.reset_depth_xforms(XformDesc.global_depth())
    s.reset_spatial_xforms(XformDesc.spatial())
    with pytest.raises(RuntimeError, match="robust_loss"):   # (refused where it is set: cvd_set_solver_options validates)
        s.set_robust_loss(5)
    s.evaluate(OptParams.defaults(), 0.1, np.zeros((3, 7)) + [0, 0, 0, 0, 0, 0, 0.2])   # the handle keeps its valid options


@pytest.mark.parametrize("cfg", ["global", "grid6x4"])
def test_huber_full_solve_reaches_the_oracle_minimum(Solver, cfg):
    v = synth.make_video(12, 96, 56, seed=1239, flow_noise_px=1.0)
    out = {}
    for k, ctor in (("hip", lambda: Solver(0)), ("oracle", Oracle)):
        s = ctor()
        synth.load_into(s, v)
        if k == "hip":
            s.set_options(pcg_relative_tolerance=1e-3, robust_loss=1)
        else:
            s.set_robust_loss(1)
        p = OptParams.defaults()
        p.num_threads = 8
        p.robustness = 0.01   # ~1 px of reprojection error at this resolution: a good part of the noisy constraints are outliers
        p.coarse_to_fine = 0
        p.num_steps = 1
        s.reset_depth_xforms(XformDesc.global_depth())
        s.reset_spatial_xforms(XformDesc.spatial())
        s.normalize_depth(p)
        if cfg != "global":
            s.grid_xform_split(XformDesc.grid_depth(6, 4))
        s.pose_optimization(p)
        out[k] = (s.get_poses(), s.get_xform_params(), s.summary())
    sh, so = out["hip"][2], out["oracle"][2]
    assert sh["termination"] == 0 and so["termination"] == 0
    assert abs(sh["final_cost"] - so["final_cost"]) <= 1e-4 * abs(so["final_cost"]), (sh["final_cost"], so["final_cost"])
    perr, rerr = synth.relative_pose_error(out["hip"][0]["position"], out["hip"][0]["orientation"],
                                           out["oracle"][0]["position"], out["oracle"][0]["orientation"])
    assert perr < 1e-2 and rerr < 1e-3, (perr, rerr)
    assert rel(out["hip"][1], out["oracle"][1]) < 1e-2
