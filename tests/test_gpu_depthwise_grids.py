"""GPU parity of DEPTH-WISE grids (XformDescriptor.gridSize.z > 1, reference lib/DepthMapTransform.cpp:709-729, 771-851): the
third grid coordinate of a sample is its source DISPARITY, (1 / d_src - 1 / depthMax) / interval; linear gather with 8 taps
(spatial x depth-wise) or 2 (depth-wise only, gridSize = (1, 1, z)); the deformation cost gains the z-edges (:996-1002).
Against the oracle's restatement (oracle/cvd_oracle.cpp: depthGather / computeGridDeformationCost)."""
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


def _desc(variant, v):
    dmin, dmax = float(v.depth[v.depth > 0].min()) * 0.9, float(v.depth.max()) * 1.1
    if variant == "grid3x2x3":
        return XformDesc.grid_depth(3, 2, depth=3, dmin=dmin, dmax=dmax)
    if variant == "grid4x3x2":
        return XformDesc.grid_depth(4, 3, depth=2, dmin=dmin, dmax=dmax)
    return XformDesc.grid_depth(1, 1, depth=4, dmin=dmin, dmax=dmax)  # depth-wise only


@pytest.mark.parametrize("variant", ["grid3x2x3", "grid4x3x2", "z_only_4"])
def test_depthwise_cost_gradient_blocks_products_and_maps(Solver, variant):
    F = 5
    v = synth.make_video(F, 96, 56, seed=91)
    objs = {"hip": Solver(0), "oracle": Oracle()}
    rng = np.random.default_rng(2)
    pose = np.zeros((F, 7))
    pose[:, :6] = rng.normal(0, 0.02, (F, 6))
    pose[:, 6] = 0.2 + rng.uniform(0, 0.02, F)
    p = OptParams.defaults()
    p.num_threads = 2
    res, maps = {}, {}
    for k, s in objs.items():
        synth.load_into(s, v)
        s.reset_depth_xforms(_desc(variant, v))
        s.reset_spatial_xforms(XformDesc.spatial())
        th = s.get_xform_params()
        s.set_xform_params(th * (1.0 + 0.1 * np.random.default_rng(9).standard_normal(th.shape)))
        res[k] = s.evaluate(p, 0.3, pose, want_gradient=True, want_hdiag=True, want_hfull=True)
        maps[k] = (s.apply_depth_xforms(), s.depth_param_maps())
    a, b = res["hip"], res["oracle"]
    assert a["num_residual_blocks"] == b["num_residual_blocks"]
    assert abs(a["cost"] - b["cost"]) <= TOL * abs(b["cost"]), (a["cost"], b["cost"])
    assert rel(a["gradient"], b["gradient"]) < TOL
    assert rel(a["hdiag"], b["hdiag"]) < TOL
    assert rel(a["hfull"], b["hfull"]) < TOL
    # DepthXform::apply / GridDepthXform::paramMap read the source depth for the third coordinate
    assert np.abs(maps["hip"][0] - maps["oracle"][0]).max() <= 1e-6 * np.abs(maps["oracle"][0]).max()
    assert np.abs(maps["hip"][1] - maps["oracle"][1]).max() <= 1e-12 * np.abs(maps["oracle"][1]).max()
    # the depth-wise axis is really in use: layers differ after the perturbation and change the transformed depth
    objs["hip"].reset_depth_xforms(_desc(variant, v))
    assert np.abs(objs["hip"].apply_depth_xforms() - maps["hip"][0]).max() > 1e-3


def test_depthwise_solve_reaches_the_oracle_minimum(Solver):
    v = synth.make_video(8, 96, 56, seed=92)
    out = {}
    for k, s in (("hip", Solver(0)), ("oracle", Oracle())):
        synth.load_into(s, v)
        p = OptParams.defaults()
        p.num_threads = 4
        p.coarse_to_fine = 0
        p.num_steps = 1
        s.reset_depth_xforms(XformDesc.global_depth())
        s.reset_spatial_xforms(XformDesc.spatial())
        s.normalize_depth(p)
        th0 = s.get_xform_params()[0, 0]
        s.reset_depth_xforms(_desc("grid3x2x3", v))
        s.set_xform_params(np.full_like(s.get_xform_params(), th0))
        s.pose_optimization(p)
        out[k] = (s.summary(), s.get_poses(), s.get_xform_params())
    fh, fo = out["hip"][0]["final_cost"], out["oracle"][0]["final_cost"]
    assert abs(fh - fo) <= 1e-5 * abs(fo), (fh, fo)
    perr, rerr = synth.relative_pose_error(out["hip"][1]["position"], out["hip"][1]["orientation"],
                                           out["oracle"][1]["position"], out["oracle"][1]["orientation"])
    assert perr < 2e-3 and rerr < 1e-3, (perr, rerr)   # (8 frames 96x56 with 18 depth handles per frame: weakly determined)


def test_depthwise_descriptor_validation(Solver):
    s = Solver(0)
    v = synth.make_video(3, 64, 40, seed=1)
    synth.load_into(s, v)
    with pytest.raises(RuntimeError, match="Depth values must be positive"):
        s.reset_depth_xforms(XformDesc.grid_depth(3, 2, depth=3, dmin=0.0, dmax=5.0))
    with pytest.raises(RuntimeError, match="Depth range must be positive"):
        s.reset_depth_xforms(XformDesc.grid_depth(3, 2, depth=3, dmin=5.0, dmax=5.0))
    with pytest.raises(RuntimeError, match="1-parameter value transforms"):
        s.reset_depth_xforms(XformDesc.grid_depth(1, 1, value=ValueXformType.ScaleShift, depth=3, dmin=1.0, dmax=5.0))
