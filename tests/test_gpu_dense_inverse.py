"""GPU (-m gpu): the dense SPD inverse of the dense coarse level (k_dense_spd_inverse, cvd_dense_inverse.h: one persistent
kernel, tiles in MFMA accumulators, one grid barrier per 16-wide pivot step) through the C ABI (cvd_dense_inverse_debug).
Known answers: numpy.linalg.inv in f64.  Sizes cover one tile, tile-boundary cases, every TPW instantiation
(S = 1 ... 12 super-tile edges), n not a multiple of 8 / 16 (identity padding) and the benchmark's 2400 (300 frames x 8).

f64 arithmetic and f64 output (an f32 copy of A_c^-1 stops being positive definite once cond(A_c) passes ~1e7, which the
damped coarse matrix does at large trust-region radii: PCG then runs into its iteration cap)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver():
    from robust_cvd_amd import api
    return api.Solver(0)


def spd(n, seed, cond=1e3):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    ev = np.exp(rng.uniform(0, np.log(cond), n))
    a = (q * ev) @ q.T
    return 0.5 * (a + a.T)


@pytest.mark.parametrize("n", [8, 16, 24, 40, 100, 192, 200, 333, 512, 1000, 1608, 2400, 3200, 4096])
def test_dense_inverse_matches_numpy(solver, n):
    a = spd(n, seed=7000 + n)
    ref = np.linalg.inv(a)
    m, failed = solver.dense_inverse_debug(a)
    assert failed == 0, failed
    assert m.dtype == np.float64
    scale = np.abs(ref).max()
    assert np.abs(m - ref).max() / scale < 1e-11, (n, np.abs(m - ref).max() / scale)
    assert np.abs(m @ a - np.eye(n)).max() < 1e-9, n
    assert np.array_equal(m, m.T)  # exactly symmetric: mirrored stores


def test_dense_inverse_layout_check(solver):
    """Distinct diagonal + a few known couplings in different tiles / super-tiles: catches a transposed or mis-ordered tile
    that a random SPD matrix could hide behind its tolerance (cdna_hip_programming.md: transpose-detecting checks)."""
    n = 400
    a = np.diag(np.arange(1, n + 1, dtype=np.float64))
    for i, j, v in ((40, 3, 0.5), (20, 17, -0.25), (399, 0, 0.75), (210, 200, 1.5), (130, 129, -0.5), (300, 77, 0.3)):
        a[i, j] = a[j, i] = v
    ref = np.linalg.inv(a)
    m, failed = solver.dense_inverse_debug(a)
    assert failed == 0
    assert np.abs(m - ref).max() < 1e-14


def test_dense_inverse_of_an_ill_conditioned_matrix_stays_positive_definite(solver):
    """cond = 1e6: the inverse as stored must still be SPD, or the two-level preconditioner built from it is indefinite.
    (An explicit inverse cannot promise that beyond cond ~ 1e8 -- its rounding errors, cond x eps relative to its LARGEST
    eigenvalue, swamp its small ones -- which is why the coarse level shifts its diagonal: coarse_dense_shift.)"""
    a = spd(480, seed=23, cond=1e6)
    m, failed = solver.dense_inverse_debug(a)
    assert failed == 0
    assert np.array_equal(m, m.T)
    assert np.linalg.eigvalsh(m).min() > 0.0
    assert np.abs(m @ a - np.eye(480)).max() < 1e-8


def test_dense_inverse_reports_a_non_positive_pivot(solver):
    n = 96
    a = spd(n, seed=5)
    a[50, 50] = -1.0
    m, failed = solver.dense_inverse_debug(a)
    assert failed == 1          # (no earlier inverse to keep: the level would be switched off)
    assert not m.any()          # the output is left untouched


def test_dense_inverse_is_repeatable(solver):
    """Tiles never move between waves and every reduction has a fixed order: two runs agree bit for bit."""
    a = spd(777, seed=11, cond=1e6)
    m0, f0 = solver.dense_inverse_debug(a)
    m1, f1 = solver.dense_inverse_debug(a)
    assert f0 == 0 and f1 == 0 and np.array_equal(m0, m1)
