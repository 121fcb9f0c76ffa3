"""GPU parity of the image operators in front of the sampler: k_bgr_to_gray / k_sobel_cov /
k_box_min_eigenval and k_chamfer_5x5 against the oracle (reference lib/FlowConstraints.cpp:257-286, 417-423).
Float work with a fixed operation order and integer chamfer arithmetic: the bar is bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair():
    from oracle.oracle import Oracle
    from robust_cvd_amd.api import Solver
    return Solver(0), Oracle()


@pytest.mark.parametrize("shape", [(3, 28, 40), (2, 224, 384), (1, 2, 2), (1, 1, 7), (2, 37, 5)])
def test_corner_min_eigenval_bit_exact(pair, shape):
    hip, orc = pair
    rng = np.random.default_rng(sum(shape))
    bgr = rng.uniform(0, 255, shape + (3,)).astype(np.float32)
    bgr[0, : shape[1] // 2, : shape[2] // 2] = 17.0   # a flat patch with a corner
    a, b = hip.corner_min_eigenval(bgr), orc.corner_min_eigenval(bgr)
    assert a.shape == shape and np.array_equal(a, b)


@pytest.mark.parametrize("shape,fill", [((3, 28, 40), 0.02), ((2, 224, 384), 0.001), ((1, 112, 192), 0.3),
                                        ((1, 5, 600), 0.05), ((1, 3, 1), 0.5)])
def test_dynamic_distance_bit_exact(pair, shape, fill):
    hip, orc = pair
    rng = np.random.default_rng(shape[2])
    mask = np.where(rng.uniform(size=shape) < fill, rng.choice(np.array([0, 126], np.uint8), size=shape),
                    rng.choice(np.array([127, 128, 255], np.uint8), size=shape)).astype(np.uint8)
    a, b = hip.dynamic_distance(mask), orc.dynamic_distance(mask)
    assert np.array_equal(a, b)
    assert np.all(a[mask < 127] == 0.0)


def test_dynamic_distance_degenerate_masks(pair):
    hip, orc = pair
    for m in (np.zeros((1, 9, 11), np.uint8), np.full((1, 9, 11), 255, np.uint8)):
        assert np.array_equal(hip.dynamic_distance(m), orc.dynamic_distance(m))


def test_image_ops_feed_the_sampler(pair):
    """corner response + dynamic distance computed on the device give the same constraints as the oracle's chain."""
    from robust_cvd_amd import synth
    hip, orc = pair
    v = synth.make_video(3, 64, 40, seed=77, spacing=9)
    rng = np.random.default_rng(5)
    bgr = rng.uniform(0, 1, (3, 40, 64, 3)).astype(np.float32)
    dyn = np.full((3, 20, 32), 255, np.uint8)
    dyn[:, 5:9, 10:20] = 0
    pairs = np.array([[0, 1], [1, 2], [2, 0]], np.int32)
    flow = rng.normal(0, 1.5, (3, 40, 64, 2)).astype(np.float32)
    mask = (rng.uniform(size=(3, 40, 64)) < 0.9).astype(np.uint8)
    res = []
    for s in (hip, orc):
        synth.load_into(s, v)
        co = s.corner_min_eigenval(bgr)
        dd = s.dynamic_distance(dyn)
        res.append(s.sample_pair_constraints(pairs, co, flow, mask, 6, dyn_dist=dd, min_dynamic_distance=2.5))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert res[0][0][-1] > 20
