"""GPU parity of the dense consumer kernels (robust_cvd_amd/csrc/cvd_dense.h): DepthXform::apply,
GridDepthXform::paramMap, SpatialXform::warp (reference lib/DepthMapTransform.cpp:394-449, 950-994) against the
oracle's per-pixel restatement.  f64 sums of <= 16 taps, stored as f32 (apply, warp) or f64 (paramMap)."""
import numpy as np
import pytest

from oracle.oracle import Oracle
from robust_cvd_amd import synth
from robust_cvd_amd.ctypes_types import SpatialXformType, ValueXformType, XformDesc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Solver():
    from robust_cvd_amd import api
    return api.Solver


CASES = {
    "global_scale": (XformDesc.global_depth(), XformDesc.spatial()),
    "grid17x10_linear": (XformDesc.grid_depth(17, 10), XformDesc.spatial(SpatialXformType.BilinearGrid, 4, 3)),
    "grid5x4_cubic_scaleshift": (XformDesc.grid_depth(5, 4, ValueXformType.ScaleShift, cubic=True),
                                 XformDesc.spatial(SpatialXformType.BicubicGrid, 4, 3)),
    "grid3x3_corners": (XformDesc.grid_depth(3, 3), XformDesc.spatial(SpatialXformType.CornersBilinear)),
    "grid3x2_vertical": (XformDesc.grid_depth(3, 2), XformDesc.spatial(SpatialXformType.VerticalLinear)),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_dense_maps_match_oracle(Solver, name):
    dd, sd = CASES[name]
    F = 5
    v = synth.make_video(F, 96, 56, seed=41, max_pairs=4)
    objs = {"hip": Solver(0), "oracle": Oracle()}
    rng = np.random.default_rng(5)
    res = {}
    for k, s in objs.items():
        synth.load_into(s, v)
        s.reset_depth_xforms(dd)
        s.reset_spatial_xforms(sd)
        th = s.get_xform_params(False)
        r1 = np.random.default_rng(1)
        s.set_xform_params(0.5 + r1.uniform(0, 1.0, th.shape), False)
        sp = s.get_xform_params(True)
        if sp.size:
            s.set_xform_params(0.05 * np.random.default_rng(2).standard_normal(sp.shape), True)
        res[k] = {"apply": s.apply_depth_xforms(1, 3), "warp": s.spatial_warp_maps(40, 70, 0, 2)}
        if name != "global_scale":
            res[k]["pmap"] = s.depth_param_maps(0, F)
    a, b = res["hip"], res["oracle"]
    assert a["apply"].shape == (3, 56, 96) and a["warp"].shape == (2, 40, 70, 2)
    # f32 storage of identical f64 sums: allow one float ulp
    np.testing.assert_allclose(a["apply"], b["apply"], rtol=2e-7, atol=0)
    np.testing.assert_allclose(a["warp"], b["warp"], rtol=2e-7, atol=1e-12)
    if "pmap" in a:
        np.testing.assert_allclose(a["pmap"], b["pmap"], rtol=1e-14, atol=0)
    else:
        with pytest.raises(RuntimeError, match="Parameter map not implemented"):
            objs["hip"].depth_param_maps(0, 1)


def test_dense_maps_full_size_properties(Solver):
    """BASELINE size (300 x 384 x 224): identity transforms reproduce the input depth bit for bit; a constant scale
    scales it; the kernel moves ~8 B per pixel."""
    F, W, H = 300, 384, 224
    rng = np.random.default_rng(3)
    s = Solver(0)
    s.set_video(F, W, H)
    depth = rng.uniform(0.5, 4.0, (F, H, W)).astype(np.float32)
    s.set_depth_all(depth)
    s.reset_poses()
    s.reset_depth_xforms(XformDesc.global_depth())
    s.reset_spatial_xforms(XformDesc.spatial())
    out, ms = s.apply_depth_xforms(timing=True)
    assert np.array_equal(out, depth)                      # theta = 1
    s.reset_depth_xforms(XformDesc.grid_depth(17, 10))
    th = s.get_xform_params(False)
    s.set_xform_params(np.full_like(th, 1.5), False)
    out, ms = s.apply_depth_xforms(timing=True)
    np.testing.assert_allclose(out, (depth.astype(np.float64) * 1.5).astype(np.float32), rtol=2e-7)
    pm = s.depth_param_maps(0, 4)
    np.testing.assert_allclose(pm, 1.5, rtol=1e-14)
    gbps = F * W * H * 8 / (ms * 1e-3) / 1e9
    print(f"k_apply_depth<4>: {ms * 1e3:.1f} us for {F}x{H}x{W}, {gbps:.0f} GB/s algorithmic")
    assert ms < 5.0
