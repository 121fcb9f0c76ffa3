"""GPU parity of k_flow_guided_filter (DepthVideoProcessor::flowGuidedFilter, reference lib/Processor.cpp:315-590)
against the oracle.  f32 in the reference's operation order; the device expf is not glibc's expf, so the bar is a
float tolerance: rtol 2e-6 on the weighted mean; the weighted median returns one of the sample depths and must be
exact except where exp's last bit moves the half-weight crossing (< 0.5 % of the pixels allowed)."""
import numpy as np
import pytest

from tests.filter_cases import make_case

pytestmark = pytest.mark.gpu

RTOL = 2e-6


@pytest.fixture(scope="module")
def pair():
    from oracle.oracle import Oracle
    from robust_cvd_amd.api import Solver
    return Solver(0), Oracle()


def _run(s, c, fr, sr, median, first=0, count=None):
    return s.flow_guided_filter(c["depth"], c["cameras"], c["flow_fwd"], c["mask_fwd"], c["flow_bwd"], c["mask_bwd"],
                                c["inv_aspect"], fr, spatial_radius=sr, median=median, first=first, count=count)


@pytest.mark.parametrize("n,w,h,fr,sr,median", [(6, 48, 28, 4, 0, False), (6, 48, 28, 4, 0, True), (5, 40, 24, 2, 1, False),
                                               (5, 40, 24, 2, 2, True), (3, 33, 17, 1, 3, True), (1, 20, 12, 4, 1, False)])
def test_filter_matches_oracle(pair, n, w, h, fr, sr, median):
    hip, orc = pair
    c = make_case(n, w, h, seed=n * 7 + sr)
    a, b = _run(hip, c, fr, sr, median), _run(orc, c, fr, sr, median)
    assert a.shape == (n, h, w)
    if median:
        assert np.mean(a == b) > 0.995
    else:
        assert np.allclose(a, b, rtol=RTOL, atol=0)


def test_filter_depth_raster_differs_from_flow_raster(pair):
    hip, orc = pair
    c = make_case(4, 36, 20, dw=72, dh=40, seed=5)
    a, b = _run(hip, c, 2, 1, False), _run(orc, c, 2, 1, False)
    assert np.allclose(a, b, rtol=RTOL, atol=0)


def test_filter_output_subrange_and_default_radius(pair):
    """The pipeline's call: spatialRadius 0, frameRadius 4 (reference params.py:212), outputs for a sub-range."""
    hip, orc = pair
    c = make_case(9, 96, 56, seed=9, flow_sigma=2.5)
    a = _run(hip, c, 4, 0, False, first=2, count=5)
    b = _run(orc, c, 4, 0, False, first=2, count=5)
    full = _run(hip, c, 4, 0, False)
    assert a.shape == (5, 56, 96) and np.allclose(a, b, rtol=RTOL, atol=0)
    assert np.array_equal(a, full[2:7])


def test_filter_masked_flow_stops_the_chain(pair):
    hip, _ = pair
    c = make_case(3, 24, 16, seed=2)
    c["mask_fwd"][:] = 0
    c["mask_bwd"][:] = 0
    a = _run(hip, c, 2, 0, False)
    assert np.allclose(a, c["depth"], rtol=1e-5)   # only the pixel's own sample is left


def test_filter_rejects_oversized_median_window(pair):
    hip, _ = pair
    c = make_case(2, 16, 12, seed=1)
    with pytest.raises(RuntimeError, match="256 samples"):
        _run(hip, c, 4, 3, True)
