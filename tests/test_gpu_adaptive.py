"""GPU parity of AdaptiveDeformationCost (reference lib/PoseOptimizer.cpp:559-656, 1449-1491): k_adaptive_weights +
the weighted deformation rows in the assembly / product / cost kernels against the oracle's dual-number restatement."""
import numpy as np
import pytest

from oracle.oracle import Oracle
from robust_cvd_amd import synth
from robust_cvd_amd.ctypes_types import OptParams, ValueXformType, XformDesc
from tests.helpers import rel

pytestmark = pytest.mark.gpu
TOL = 1e-9


@pytest.fixture(scope="module")
def Solver():
    from robust_cvd_amd import api
    return api.Solver


def _masks(F, seed=4):
    rng = np.random.default_rng(seed)
    m = np.where(rng.uniform(size=(F, 20, 32)) < 0.3, 0, 255).astype(np.uint8)
    m[:, 5:12, 8:20] = 0
    return m


@pytest.mark.parametrize("value,cubic", [(ValueXformType.Scale, False), (ValueXformType.Scale, True),
                                         (ValueXformType.ScaleShift, True)])
def test_adaptive_cost_gradient_hessian_match_oracle(Solver, value, cubic):
    F, gw, gh = 4, 5, 4
    v = synth.make_video(F, 64, 40, seed=62, spacing=9)
    masks = _masks(F)
    rng = np.random.default_rng(7)
    N = 2 if value == ValueXformType.ScaleShift else 1
    theta = 1.0 + 0.15 * rng.standard_normal((F, gw * gh * N))
    pose = np.zeros((F, 7)); pose[:, :6] = rng.normal(0, 0.03, (F, 6)); pose[:, 6] = 0.2
    p = OptParams.defaults()
    p.adaptive_deformation_cost = 3.0
    res = {}
    for k, s in {"hip": Solver(0), "oracle": Oracle()}.items():
        synth.load_into(s, v)
        s.reset_depth_xforms(XformDesc.grid_depth(gw, gh, value, cubic=cubic))
        s.reset_spatial_xforms(XformDesc.spatial())
        s.set_xform_params(theta)
        if k == "hip":
            with pytest.raises(RuntimeError, match="requires a dynamic mask stream"):
                s.evaluate(p, 0.4, pose)
        s.set_dynamic_masks(masks)
        res[k] = s.evaluate(p, 0.4, pose, want_hdiag=True, want_hfull=True)
        p0 = OptParams.defaults()
        res[k + "_plain"] = s.evaluate(p0, 0.4, pose)["cost"]
    assert abs(res["hip"]["cost"] - res["oracle"]["cost"]) < TOL * abs(res["oracle"]["cost"])
    assert abs(res["oracle"]["cost"] - res["oracle_plain"]) > 1e-3       # the adaptive term is active
    assert rel(res["hip"]["gradient"], res["oracle"]["gradient"]) < TOL
    assert rel(res["hip"]["hdiag"], res["oracle"]["hdiag"]) < TOL
    assert rel(res["hip"]["hfull"], res["oracle"]["hfull"]) < TOL


def test_adaptive_full_solve_reaches_the_oracle_minimum(Solver):
    F = 5
    v = synth.make_video(F, 64, 40, seed=63, spacing=9)
    masks = _masks(F, seed=9)
    p = OptParams.defaults()
    p.adaptive_deformation_cost = 2.0
    p.num_threads = 2
    p.ctf_long, p.ctf_short = 6, 4
    out = {}
    for k, s in {"hip": Solver(0), "oracle": Oracle()}.items():
        synth.load_into(s, v)
        s.set_dynamic_masks(masks)
        s.reset_depth_xforms(XformDesc.global_depth())
        s.reset_spatial_xforms(XformDesc.spatial())
        s.normalize_depth(p)
        s.pose_optimization(p)
        out[k] = (s.summary()["final_cost"], s.get_xform_params())
    assert abs(out["hip"][0] - out["oracle"][0]) < 1e-5 * out["oracle"][0]
    assert rel(out["hip"][1], out["oracle"][1]) < 5e-3
    # Global transforms have no deformation residuals: adaptive is silently skipped there (reference :1465-1467)
    s = Solver(0)
    synth.load_into(s, v)
    s.reset_depth_xforms(XformDesc.global_depth())
    s.reset_spatial_xforms(XformDesc.spatial())
    s.evaluate(p, 0.4)
