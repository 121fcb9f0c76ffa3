"""Known-answer tests pinning the oracle's building blocks (the reference ships no tests: SURVEY.md 4).

Every expectation is derived independently of oracle/cvd_oracle.cpp: hand-computed values, closed forms,
or a separate numpy restatement of the reference source lines cited in each test.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from robust_cvd_amd.ctypes_types import SpatialXformType, ValueXformType, XformDesc


# ---- independent numpy restatement of the gathers (reference lib/DepthMapTransform.cpp:671-678, 750-764,
# 823-840, 911-947) ---------------------------------------------------------------------------------------
def cell(loc, g):
    maxc = np.nextafter(np.float64(g - 1), 0.0)
    s = min(max((np.float64(np.float32(loc)) + 1.0) * (g - 1) / 2.0, 0.0), maxc)
    i = int(s)
    return i, s - i


def catmull(t):
    return np.array([-0.5 * t**3 + t**2 - 0.5 * t, 1.5 * t**3 - 2.5 * t**2 + 1.0,
                     -1.5 * t**3 + 2.0 * t**2 + 0.5 * t, 0.5 * t**3 - 0.5 * t**2])


def ref_bilinear(lx, ly, gx, gy):
    ix, rx = cell(lx, gx)
    iy, ry = cell(ly, gy)
    i0 = ix + iy * gx
    return ([i0, i0 + 1, i0 + gx, i0 + gx + 1],
            [(1 - rx) * (1 - ry), rx * (1 - ry), (1 - rx) * ry, rx * ry])


def ref_bicubic(lx, ly, gx, gy):
    ix, rx = cell(lx, gx)
    iy, ry = cell(ly, gy)
    wx, wy = catmull(rx), catmull(ry)
    acc = {}
    order = []
    for y in range(4):
        for x in range(4):
            px = min(max(ix - 1 + x, 0), gx - 1)   # out-of-range taps fold onto the clamped neighbour
            py = min(max(iy - 1 + y, 0), gy - 1)
            key = px + py * gx
            if key not in acc:
                acc[key] = 0.0
                order.append(key)
            acc[key] += wx[x] * wy[y]
    order.sort()
    return order, [acc[k] for k in order]


GRIDS = [(4, 4), (17, 10), (16, 12), (2, 2), (6, 4), (3, 5)]
LOCS = [(-1.0, -1.0), (1.0, 1.0), (0.0, 0.0), (-0.999, 0.37), (0.61, -0.83), (0.999999, -0.5), (0.25, 1.0)]


@pytest.mark.parametrize("gx,gy", GRIDS)
def test_bilinear_depth_gather(gx, gy):
    d = XformDesc.grid_depth(gx, gy)
    for lx, ly in LOCS:
        idx, w = orc.gather(d, 2.0, lx, ly)
        ridx, rw = ref_bilinear(lx, ly, gx, gy)
        assert list(idx) == ridx
        np.testing.assert_allclose(w, rw, rtol=0, atol=1e-15)
        assert abs(w.sum() - 1.0) < 1e-14 and (w >= 0).all()


@pytest.mark.parametrize("gx,gy", GRIDS)
def test_bicubic_gather_with_border_folding(gx, gy):
    dd = XformDesc.grid_depth(gx, gy, cubic=True)
    ds = XformDesc.spatial(SpatialXformType.BicubicGrid, gx, gy)
    for lx, ly in LOCS:
        ridx, rw = ref_bicubic(lx, ly, gx, gy)
        for desc in (dd, ds):
            idx, w = orc.gather(desc, 2.0, lx, ly)
            assert list(idx) == ridx, (gx, gy, lx, ly)
            np.testing.assert_allclose(w, rw, rtol=0, atol=2e-15)
            assert abs(w.sum() - 1.0) < 1e-13
        n_expected = len(set(min(max(cell(lx, gx)[0] - 1 + x, 0), gx - 1) for x in range(4))) * \
            len(set(min(max(cell(ly, gy)[0] - 1 + y, 0), gy - 1) for y in range(4)))
        assert len(ridx) == n_expected


def test_cubic_interior_is_catmull_rom_and_interpolates_vertices():
    d = XformDesc.grid_depth(8, 8, cubic=True)
    # at a vertex the spline interpolates: weight 1 on that vertex
    lx = -1 + 2 * 3 / 7.0
    ly = -1 + 2 * 4 / 7.0
    idx, w = orc.gather(d, 1.0, lx, ly)
    k = int(np.argmax(w))
    assert idx[k] == 3 + 4 * 8 and abs(w[k] - 1) < 1e-6 and np.abs(np.delete(w, k)).max() < 1e-6
    # hand value: t = 0.5 -> taps (-1/16, 9/16, 9/16, -1/16)
    np.testing.assert_allclose(catmull(0.5), [-0.0625, 0.5625, 0.5625, -0.0625])
    lxm = -1 + 2 * 3.5 / 7.0
    idx, w = orc.gather(d, 1.0, lxm, ly)
    assert len(idx) == 16
    row = w.reshape(4, 4).sum(0)
    np.testing.assert_allclose(row, [-0.0625, 0.5625, 0.5625, -0.0625], atol=1e-6)


def test_grid_row_zero_is_image_bottom():
    """ndc.y = +1 (image top) maps to the LAST grid row."""
    d = XformDesc.grid_depth(3, 3)
    idx, w = orc.gather(d, 1.0, -1.0, 1.0)
    assert idx[np.argmax(w)] in (6, 3)  # clamped just below row 2: weight mostly on vertex (0, 2) = 6
    assert idx[np.argmax(w)] == 6
    idx, w = orc.gather(d, 1.0, -1.0, -1.0)
    assert idx[np.argmax(w)] == 0


def test_global_and_identity_gathers():
    idx, w = orc.gather(XformDesc.global_depth(), 3.0, 0.3, -0.2)
    assert list(idx) == [0] and list(w) == [1.0]
    idx, w = orc.gather(XformDesc.identity_depth(), 3.0, 0.3, -0.2)
    assert len(idx) == 0


def test_small_spatial_gathers():
    # reference lib/DepthMapTransform.cpp:1109-1113 and :1183-1189
    idx, w = orc.gather(XformDesc.spatial(SpatialXformType.VerticalLinear), 0, 0.3, 0.5)
    assert list(idx) == [0, 1]
    np.testing.assert_allclose(w, [0.75, 0.25])
    idx, w = orc.gather(XformDesc.spatial(SpatialXformType.CornersBilinear), 0, 0.5, -0.5)
    np.testing.assert_allclose(w, [0.75 * 0.25, 0.25 * 0.25, 0.75 * 0.75, 0.25 * 0.75])
    idx, w = orc.gather(XformDesc.spatial(SpatialXformType.BilinearGrid, 4, 3), 0, 0.1, 0.2)
    ridx, rw = ref_bilinear(0.1, 0.2, 4, 3)
    assert list(idx) == ridx
    np.testing.assert_allclose(w, rw, atol=1e-15)


def test_linear_grid_with_two_value_params_is_rejected():
    """reference lib/DepthMapTransform.cpp:829 indexes &params_[i] (not i*N): aliasing blocks, undefined."""
    with pytest.raises(RuntimeError):
        orc.gather(XformDesc.grid_depth(4, 4, ValueXformType.ScaleShift), 1.0, 0.0, 0.0)


# ---- rotations (ceres/rotation.h + Eigen semantics) ---------------------------------------------------------
def rodrigues(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th == 0:
        return np.eye(3)
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th**2 * K @ K


def test_angle_axis_rotate_point_matches_rodrigues():
    rng = np.random.default_rng(0)
    for _ in range(20):
        w = rng.normal(0, 1.0, 3)
        p = rng.normal(0, 2.0, 3)
        np.testing.assert_allclose(orc.angle_axis_rotate_point(w, p), rodrigues(w) @ p, atol=1e-14)


def test_angle_axis_small_angle_branch_is_first_order():
    w = np.array([1e-9, -2e-9, 3e-9])   # theta^2 < DBL_EPSILON
    p = np.array([0.3, -1.2, 2.0])
    np.testing.assert_array_equal(orc.angle_axis_rotate_point(w, p), p + np.cross(w, p))
    np.testing.assert_array_equal(orc.angle_axis_rotate_point(np.zeros(3), p), p)


def test_rotation_conversions_round_trip():
    rng = np.random.default_rng(1)
    for _ in range(20):
        w = rng.normal(0, 0.8, 3)
        R = orc.angle_axis_to_rotation_matrix(w)
        np.testing.assert_allclose(R, rodrigues(w), atol=1e-14)
        np.testing.assert_allclose(orc.rotation_matrix_to_angle_axis(R), w, atol=1e-12)
        q = orc.rotation_matrix_to_quaternion(R)  # x, y, z, w
        th = np.linalg.norm(w)
        expect = np.r_[np.sin(th / 2) * w / th, np.cos(th / 2)]
        if np.dot(q, expect) < 0:   # q and -q are the same rotation (Eigen's trace <= 0 branch)
            expect = -expect
        np.testing.assert_allclose(q, expect, atol=1e-13)
    np.testing.assert_array_equal(orc.rotation_matrix_to_angle_axis(np.eye(3)), np.zeros(3))
    # trace < 0 branch
    w = np.array([0.0, 3.0, 0.0])
    R = rodrigues(w)
    np.testing.assert_allclose(orc.rotation_matrix_to_angle_axis(R), w, atol=1e-12)
    q = orc.rotation_matrix_to_quaternion(R)
    np.testing.assert_allclose(q, [0, np.sin(1.5), 0, np.cos(1.5)], atol=1e-13)


# ---- deformation cost (reference lib/DepthMapTransform.cpp:631-667) ---------------------------------------
def test_grid_deformation_cost_values_and_order():
    d = XformDesc.grid_depth(3, 2)
    p = np.array([1.0, 2.0, -4.0, 0.5, 2.0, 3.0])  # row-major, x fastest
    r = orc.deformation_cost(d, p)
    # vertex order (y, x); per vertex: x-1 neighbour then y-1 neighbour; (this - that) / min(|this|, |that|)
    exp = [(2 - 1) / 1.0, (-4 - 2) / 2.0,                       # row 0: x = 1, 2
           (0.5 - 1) / 0.5,                                     # (0,1): y-1
           (2 - 0.5) / 0.5, (2 - 2) / 2.0,                      # (1,1): x-1, y-1
           (3 - 2) / 2.0, (3 - -4) / 3.0]                       # (2,1): x-1, y-1
    np.testing.assert_allclose(r, exp)
    assert len(r) == (3 - 1) * 2 + 3 * (2 - 1)


def test_spatial_deformation_cost_is_the_parameters():
    d = XformDesc.spatial(SpatialXformType.BilinearGrid, 3, 2)
    p = np.arange(12, dtype=np.float64) * 0.1
    np.testing.assert_array_equal(orc.deformation_cost(d, p), p)
