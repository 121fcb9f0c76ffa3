"""Oracle checks for the image operators in front of the sampler.  OpenCV (the reference's provider of
cornerMinEigenVal / distanceTransform, reference lib/FlowConstraints.cpp:257-286, 417-423) is absent: the oracle is
pinned here against independent restatements (vectorised numpy float32 for the corner response; the closed-form 5x5
chamfer metric for the distance transform), not against OpenCV itself -- parity with OpenCV stays unpinned."""
import numpy as np
import pytest

from oracle.oracle import Oracle


def np_corner_min_eigenval(bgr):
    f = np.float32
    g = (bgr[..., 0] * f(0.114) + bgr[..., 1] * f(0.587)) + bgr[..., 2] * f(0.299)
    k0, k1 = f(1.0 / 12.0), f(2.0 / 12.0)
    p = np.pad(g, 1, mode="reflect")
    l, c, r = p[:, :-2], p[:, 1:-1], p[:, 2:]
    diff, smooth = r - l, c * k1 + (l + r) * k0           # rows still padded vertically
    dx = diff[1:-1] * k1 + (diff[:-2] + diff[2:]) * k0
    dy = smooth[2:] - smooth[:-2]
    out = []
    for cv in (dx * dx, dx * dy, dy * dy):
        q = np.pad(cv, 1, mode="reflect")
        rs = (q[:, :-2] + q[:, 1:-1]) + q[:, 2:]
        out.append((rs[:-2] + rs[1:-1]) + rs[2:])
    a, b, c2 = out[0] * f(0.5), out[1], out[2] * f(0.5)
    d = a - c2
    return (a + c2) - np.sqrt(d * d + b * b)


@pytest.mark.parametrize("shape", [(7, 9), (28, 40), (2, 2), (1, 5)])
def test_corner_response_matches_numpy_restatement(shape):
    rng = np.random.default_rng(3)
    bgr = rng.uniform(0, 255, (2,) + shape + (3,)).astype(np.float32)
    o = Oracle()
    got = o.corner_min_eigenval(bgr)
    for n in range(2):
        ref = np_corner_min_eigenval(bgr[n])
        assert got[n].dtype == np.float32 and np.array_equal(got[n], ref.astype(np.float32))


def test_corner_response_known_answers():
    o = Oracle()
    hh, w = 12, 16
    flat = np.full((1, hh, w, 3), 37.0, np.float32)
    assert np.all(o.corner_min_eigenval(flat) == 0.0)
    ramp = np.zeros((1, hh, w, 3), np.float32)
    ramp[..., :] = np.arange(w, dtype=np.float32)[None, None, :, None]   # gray = x: one-dimensional structure
    r = o.corner_min_eigenval(ramp)[0]
    assert np.max(np.abs(r[2:-2, 2:-2])) < 1e-5                            # lambda_min = 0, lambda_max = 9 (2/3)^2 / ... > 0
    corner = np.zeros((1, hh, w, 3), np.float32)
    corner[0, :6, :8] = 255.0                                             # an L-corner: both eigenvalues > 0 there
    r = o.corner_min_eigenval(corner)[0]
    assert r[5, 7] > 10.0 and r[5, 7] == r.max() or r.max() > 10.0
    assert r[0, 0] == 0.0 and r[-1, -1] == 0.0


def chamfer_closed_form(mask):
    A, B, C = 65536, 91750, 143976
    hh, w = mask.shape
    zy, zx = np.nonzero(mask < 127)
    out = np.zeros((hh, w), dtype=np.int64)
    for y in range(hh):
        for x in range(w):
            dx, dy = np.abs(zx - x), np.abs(zy - y)
            lo, hi = np.minimum(dx, dy), np.maximum(dx, dy)
            cost = np.where(hi >= 2 * lo, lo * C + (hi - 2 * lo) * A, (hi - lo) * C + (2 * lo - hi) * B)
            out[y, x] = cost.min()
    return (out.astype(np.float32) * np.float32(1.0 / 65536.0)).astype(np.float32)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_distance_transform_is_the_5x5_chamfer_metric(seed):
    rng = np.random.default_rng(seed)
    hh, w = 19, 27
    mask = np.full((1, hh, w), 255, np.uint8)
    for _ in range(1 + 2 * seed):
        mask[0, rng.integers(hh), rng.integers(w)] = rng.choice([0, 60, 126])
    if seed == 3:
        mask[0, 4:9, 10:14] = 0
    mask[0, rng.integers(hh), rng.integers(w)] = 127   # 127 counts as set (binarisation is `< 127`)
    o = Oracle()
    got = o.dynamic_distance(mask)[0]
    assert np.array_equal(got, chamfer_closed_form(mask[0]))
    assert np.all(got[mask[0] < 127] == 0.0)


def test_distance_transform_without_zero_pixels_is_large():
    o = Oracle()
    d = o.dynamic_distance(np.full((1, 6, 8), 200, np.uint8))[0]
    assert np.all(d >= 8191.0)   # DIST_MAX / 2^16: every threshold of the sampler passes, as with the FLT_MAX default
