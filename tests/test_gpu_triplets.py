"""GPU parity of the scene-flow smoothness (triplet) loss, robust_cvd_amd/csrc/cvd_triplets.h, against the oracle's
dual-number restatement of SceneFlowSmoothnessLoss (reference lib/PoseOptimizer.cpp:321-423, 1242-1339)."""
import numpy as np
import pytest

from oracle.oracle import Oracle
from robust_cvd_amd import synth
from robust_cvd_amd.ctypes_types import (IntrinsicsOptimization, OptParams, SmoothLossType, SpatialXformType, XformDesc)
from tests.helpers import rel

pytestmark = pytest.mark.gpu
TOL = 1e-9


@pytest.fixture(scope="module")
def Solver():
    from robust_cvd_amd import api
    return api.Solver


def _pair(Solver, v, trip):
    objs = {"hip": Solver(0), "oracle": Oracle()}
    for s in objs.values():
        synth.load_into(s, v)
        s.set_triplet_constraints(*trip)
    return objs


def _params(smooth_type, ws=2.0, wd=0.5):
    p = OptParams.defaults()
    p.num_threads = 2
    p.smooth_loss_type = smooth_type
    p.smooth_static_weight = ws
    p.smooth_dynamic_weight = wd
    return p


@pytest.mark.parametrize("smooth_type", [SmoothLossType.ReproDisparityLaplacian, SmoothLossType.EuclideanLaplacian,
                                         SmoothLossType.ReproDepthRatioConsistency,
                                         SmoothLossType.ReproLogDepthConsistency])
@pytest.mark.parametrize("variant", ["grid3x2", "global_bilinear_spatial", "cubic4x4_scaleshift"])
def test_triplet_cost_gradient_hessian_match_oracle(Solver, smooth_type, variant):
    F = 6
    v = synth.make_video(F, 96, 56, seed=5)
    trip = synth.make_triplets(v, spacing=20.0)
    objs = _pair(Solver, v, trip)
    rng = np.random.default_rng(3)
    pose = np.zeros((F, 7))
    pose[:, :6] = rng.normal(0, 0.03, (F, 6))
    pose[:, 6] = 0.2 + rng.uniform(0, 0.02, F)
    p = _params(smooth_type)
    res = {}
    for k, s in objs.items():
        if variant == "grid3x2":
            s.reset_depth_xforms(XformDesc.grid_depth(3, 2))
            s.reset_spatial_xforms(XformDesc.spatial())
        elif variant == "global_bilinear_spatial":
            s.reset_depth_xforms(XformDesc.global_depth())
            s.reset_spatial_xforms(XformDesc.spatial(SpatialXformType.BilinearGrid, 3, 2))
        else:
            from robust_cvd_amd.ctypes_types import ValueXformType
            s.reset_depth_xforms(XformDesc.grid_depth(4, 4, value=ValueXformType.ScaleShift, cubic=True))
            s.reset_spatial_xforms(XformDesc.spatial())
        th = s.get_xform_params(False)
        r2 = np.random.default_rng(9)
        s.set_xform_params(th * (1.0 + 0.05 * r2.standard_normal(th.shape)) if variant != "cubic4x4_scaleshift"
                           else th + 0.02 * r2.standard_normal(th.shape), False)
        sp = s.get_xform_params(True)
        if sp.size:
            s.set_xform_params(0.01 * np.random.default_rng(4).standard_normal(sp.shape), True)
        res[k] = s.evaluate(p, 0.1, pose, want_hdiag=True, want_hfull=True)
    a, b = res["hip"], res["oracle"]
    assert a["num_residual_blocks"] == b["num_residual_blocks"]
    assert abs(a["cost"] - b["cost"]) <= TOL * abs(b["cost"])
    assert rel(a["gradient"], b["gradient"]) < TOL
    assert rel(a["hdiag"], b["hdiag"]) < TOL
    assert rel(a["hfull"], b["hfull"]) < TOL
    # the loss really is on: without the weights the cost differs
    p0 = _params(smooth_type, 0.0, 0.0)
    assert abs(objs["hip"].evaluate(p0, 0.1, pose)["cost"] - a["cost"]) > 1e-3 * abs(a["cost"])


def test_triplet_weights_follow_the_static_flag(Solver):
    """ScaledLoss(w): static constraints take smoothStaticWeight, the others smoothDynamicWeight (reference :1330)."""
    F = 5
    v = synth.make_video(F, 96, 56, seed=8)
    ce, off, loc6, st = synth.make_triplets(v, spacing=20.0, dynamic_fraction=0.5)
    objs = _pair(Solver, v, (ce, off, loc6, st))
    pose = np.zeros((F, 7)); pose[:, 6] = 0.2
    costs = {}
    for ws, wd in ((1.0, 0.0), (0.0, 1.0), (1.0, 1.0), (0.0, 0.0)):
        p = _params(SmoothLossType.ReproDisparityLaplacian, ws, wd)
        for k, s in objs.items():
            s.reset_depth_xforms(XformDesc.global_depth()); s.reset_spatial_xforms(XformDesc.spatial())
            costs[(k, ws, wd)] = s.evaluate(p, 0.0, pose)["cost"] if (ws > 0 or wd > 0) else \
                s.evaluate(_params(SmoothLossType.ReproDisparityLaplacian, 0.0, 0.0), 0.0, pose)["cost"]
    base = costs[("hip", 0.0, 0.0)]
    for key in ((1.0, 0.0), (0.0, 1.0), (1.0, 1.0)):
        assert abs(costs[("hip",) + key] - costs[("oracle",) + key]) <= TOL * abs(costs[("oracle",) + key])
    # additivity of the two weight classes
    assert abs((costs[("hip", 1.0, 0.0)] - base) + (costs[("hip", 0.0, 1.0)] - base) - (costs[("hip", 1.0, 1.0)] - base)) \
        <= 1e-9 * abs(costs[("hip", 1.0, 1.0)])


def test_triplets_frame_range_and_missing_groups(Solver):
    F = 8
    v = synth.make_video(F, 96, 56, seed=12)
    ce, off, loc6, st = synth.make_triplets(v, spacing=20.0)
    objs = _pair(Solver, v, (ce, off, loc6, st))
    pose = np.zeros((F, 7)); pose[:, 6] = 0.2
    p = _params(SmoothLossType.ReproDisparityLaplacian)
    p.set_frame_range([0, 1, 2, 3, 5, 6, 7])       # triples (3,4,5), (2,3,4), (4,5,6) do not exist with frame 4 out
    res = {}
    for k, s in objs.items():
        s.reset_depth_xforms(XformDesc.grid_depth(3, 2)); s.reset_spatial_xforms(XformDesc.spatial())
        res[k] = s.evaluate(p, 0.1, pose, want_hdiag=True)
    assert res["hip"]["num_residual_blocks"] == res["oracle"]["num_residual_blocks"]
    assert abs(res["hip"]["cost"] - res["oracle"]["cost"]) <= TOL * abs(res["oracle"]["cost"])
    assert rel(res["hip"]["gradient"], res["oracle"]["gradient"]) < TOL
    assert rel(res["hip"]["hdiag"], res["oracle"]["hdiag"]) < TOL
    # a missing group for an in-range triple is an error in the reference (lib/PoseOptimizer.cpp:1260-1262)
    keep = ce != 2
    idx = np.concatenate([np.arange(off[i], off[i + 1]) for i in range(len(ce)) if keep[i]])
    off2 = np.concatenate([[0], np.cumsum(np.diff(off)[keep])]).astype(np.int64)
    s = Solver(0)
    synth.load_into(s, v)
    s.set_triplet_constraints(ce[keep], off2, loc6[idx], st[idx])
    s.reset_depth_xforms(XformDesc.global_depth()); s.reset_spatial_xforms(XformDesc.spatial())
    with pytest.raises(RuntimeError, match="Missing triplet constraints"):
        s.evaluate(_params(SmoothLossType.ReproDisparityLaplacian), 0.0, pose)


def test_full_solve_with_smoothness_reaches_the_oracle_minimum(Solver):
    F = 10
    v = synth.make_video(F, 96, 56, seed=21)
    trip = synth.make_triplets(v, spacing=16.0)
    objs = _pair(Solver, v, trip)
    out = {}
    for k, s in objs.items():
        s.reset_depth_xforms(XformDesc.global_depth()); s.reset_spatial_xforms(XformDesc.spatial())
        p = _params(SmoothLossType.ReproDisparityLaplacian, 1.0, 0.25)
        p.num_threads = 4
        p.ctf_long, p.ctf_short = 6, 4
        s.normalize_depth(p)
        s.pose_optimization(p)
        out[k] = s.summary()
    assert abs(out["hip"]["final_cost"] - out["oracle"]["final_cost"]) <= 1e-4 * abs(out["oracle"]["final_cost"])


def test_unsupported_triplet_configurations_fail_loudly(Solver):
    v = synth.make_video(5, 96, 56, seed=2)
    s = Solver(0)
    synth.load_into(s, v)
    s.set_triplet_constraints(*synth.make_triplets(v, spacing=20.0))
    s.reset_depth_xforms(XformDesc.global_depth()); s.reset_spatial_xforms(XformDesc.spatial())
    pose = np.zeros((5, 7)); pose[:, 6] = 0.2
    p = _params(SmoothLossType.ReproDepthRatioConsistency)
    p.smooth_loss_type = 7
    with pytest.raises(RuntimeError, match="Invalid loss type"):     # reference lib/PoseOptimizer.cpp:407
        s.evaluate(p, 0.0, pose)


@pytest.mark.parametrize("smooth_type", [SmoothLossType.ReproDisparityLaplacian, SmoothLossType.EuclideanLaplacian])
@pytest.mark.parametrize("variant", ["grid3x2", "global"])
def test_triplets_with_shared_intrinsics_match_oracle(Solver, smooth_type, variant):
    """IntrinsicsOptimization::Shared with the smoothness loss (reference lib/PoseOptimizer.cpp:1306-1330): the focal
    block of all three observations is frame 0's."""
    F = 6
    v = synth.make_video(F, 96, 56, seed=6)
    trip = synth.make_triplets(v, spacing=20.0)
    objs = _pair(Solver, v, trip)
    rng = np.random.default_rng(4)
    pose = np.zeros((F, 7))
    pose[:, :6] = rng.normal(0, 0.03, (F, 6))
    pose[:, 6] = 0.21
    p = _params(smooth_type)
    p.intr_opt = IntrinsicsOptimization.Shared
    res = {}
    for k, s in objs.items():
        s.reset_depth_xforms(XformDesc.grid_depth(3, 2) if variant == "grid3x2" else XformDesc.global_depth())
        s.reset_spatial_xforms(XformDesc.spatial())
        th = s.get_xform_params(False)
        s.set_xform_params(th * (1.0 + 0.05 * np.random.default_rng(9).standard_normal(th.shape)), False)
        res[k] = s.evaluate(p, 0.1, pose, want_hdiag=True, want_hfull=True)
    a, b = res["hip"], res["oracle"]
    assert a["num_residual_blocks"] == b["num_residual_blocks"]
    assert abs(a["cost"] - b["cost"]) <= TOL * abs(b["cost"])
    assert rel(a["gradient"], b["gradient"]) < TOL
    assert rel(a["hdiag"], b["hdiag"]) < TOL
    assert rel(a["hfull"], b["hfull"]) < TOL
    # one focal length: the constraints' focal column is frame 0's slot (the other frames' slots only see their own
    # focal regulariser, reference lib/PoseOptimizer.cpp:1524-1549)
    assert np.ptp(a["gradient"][1:, 6]) < 1e-12 and abs(a["gradient"][0, 6] - a["gradient"][1, 6]) > 1e-6
    p0 = _params(smooth_type, 0.0, 0.0)
    p0.intr_opt = IntrinsicsOptimization.Shared
    assert abs(objs["hip"].evaluate(p0, 0.1, pose)["cost"] - a["cost"]) > 1e-3 * abs(a["cost"])


def test_full_solve_with_smoothness_and_shared_intrinsics(Solver):
    F = 10
    v = synth.make_video(F, 96, 56, seed=22)
    trip = synth.make_triplets(v, spacing=16.0)
    objs = _pair(Solver, v, trip)
    out = {}
    for k, s in objs.items():
        s.reset_depth_xforms(XformDesc.global_depth()); s.reset_spatial_xforms(XformDesc.spatial())
        p = _params(SmoothLossType.ReproDisparityLaplacian, 1.0, 0.25)
        p.intr_opt = IntrinsicsOptimization.Shared
        p.num_threads = 4
        p.ctf_long, p.ctf_short = 6, 4
        s.normalize_depth(p)
        s.pose_optimization(p)
        out[k] = (s.summary(), s.get_poses())
    assert abs(out["hip"][0]["final_cost"] - out["oracle"][0]["final_cost"]) <= 1e-5 * abs(out["oracle"][0]["final_cost"])
    assert np.abs(out["hip"][1]["vfov"] - out["oracle"][1]["vfov"]).max() < 5e-4   # (10 frames 96x56: weakly determined)
    assert np.ptp(out["hip"][1]["vfov"]) == 0.0   # one field of view for all frames (reference :979-982)
