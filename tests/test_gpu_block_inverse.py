"""GPU (-m gpu): the per-frame dense solve of the block-Jacobi preconditioner, (H_ff + diag(lam))^-1, through the C ABI
(cvd_block_inverse_debug).  Known answers: numpy.linalg.inv in f64; the three device variants (blocked sweep on the f64
matrix cores = the default path, scalar register-tile sweep, LDS Cholesky) must also agree with each other.

Tolerance: the kernels compute in f64 and store f32, so |M A - I| is bounded by the f32 rounding of M times cond(A):
blocks are built with cond ~ 1e3, bar 2e-3 (observed ~1e-4); in f32 units of the inverse itself the bar is 4 ulp-ish
(2e-6 relative to |A^-1|_max)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver():
    from robust_cvd_amd import api
    return api.Solver(0)


def spd_blocks(n, B, seed, cond=1e3):
    rng = np.random.default_rng(seed)
    out = np.zeros((n, B, B))
    for k in range(n):
        q, _ = np.linalg.qr(rng.normal(size=(B, B)))
        ev = np.exp(rng.uniform(0, np.log(cond), B))
        a = (q * ev) @ q.T
        out[k] = 0.5 * (a + a.T)
    return out


# block sizes of the coarse-to-fine schedule (8, 31, 91, 177), configs[4] (199), tile-boundary cases, the largest block of the
# register-resident kernels (256) and blocks beyond it (257, 347 = ScaleShift on 17x10, 512), which take the batched rocSOLVER route
SIZES = [1, 7, 8, 15, 16, 17, 31, 32, 33, 91, 96, 128, 177, 192, 199, 208, 209, 255, 256, 257, 347, 512]


@pytest.mark.parametrize("B", SIZES)
def test_block_inverse_matches_numpy(solver, B):
    a = spd_blocks(5, B, seed=100 + B)
    ref = np.linalg.inv(a)
    # (the LDS Cholesky keeps the packed triangle in LDS: B <= 199; beyond 256 every variant is the batched rocSOLVER route)
    for variant in ((0, 1, 2) if B <= 199 else ((0, 1) if B <= 256 else (0,))):
        m, failed = solver.block_inverse_debug(a, variant)
        assert failed == 0, (variant, failed)
        m = m.astype(np.float64)
        scale = np.abs(ref).max(axis=(1, 2), keepdims=True)
        assert (np.abs(m - ref) / scale).max() < 2e-6, (variant, B, (np.abs(m - ref) / scale).max())
        assert np.abs(m @ a - np.eye(B)).max() < 2e-3, (variant, B)
        assert np.array_equal(m, m.transpose(0, 2, 1))     # exactly symmetric: mirrored stores


def test_block_inverse_asymmetric_layout_check(solver):
    """A block-diagonal matrix with DISTINCT diagonal blocks and one known off-diagonal coupling: catches a transposed or
    mis-ordered tile in the accumulator layout that a random SPD matrix could hide behind its tolerance."""
    B = 48
    a = np.zeros((1, B, B))
    a[0] = np.diag(np.arange(1, B + 1, dtype=np.float64))
    a[0, 40, 3] = a[0, 3, 40] = 0.5      # couples tile (2, 0) only
    a[0, 20, 17] = a[0, 17, 20] = -0.25  # inside tile (1, 1)
    ref = np.linalg.inv(a[0])
    for variant in (0, 1, 2):
        m, failed = solver.block_inverse_debug(a, variant)
        assert failed == 0
        assert np.abs(m[0] - ref).max() < 1e-6, variant
        nz = np.abs(ref) > 1e-12
        assert np.array_equal(np.abs(m[0]) > 1e-9, nz), variant   # same sparsity pattern: nothing lands in a wrong slot


def test_block_inverse_reports_non_positive_pivots(solver):
    a = spd_blocks(3, 40, seed=7)
    a[1] = -a[1]
    for variant in (0, 1):
        _, failed = solver.block_inverse_debug(a, variant)
        assert failed > 0, variant
    a = spd_blocks(3, 300, seed=8)   # (the batched route: the failed block is reported, the others are inverted)
    a[2] = -a[2]
    m, failed = solver.block_inverse_debug(a, 0)
    assert failed == 1
    assert np.abs(m[0].astype(np.float64) @ a[0] - np.eye(300)).max() < 2e-3


def test_many_blocks_full_size(solver):
    """BASELINE configs[2] shape: 300 blocks of 177; every block checked through a residual (size-independent property)."""
    a = spd_blocks(300, 177, seed=5, cond=1e2)
    m, failed = solver.block_inverse_debug(a, 0)
    assert failed == 0
    r = np.einsum("nij,njk->nik", m.astype(np.float64), a) - np.eye(177)
    assert np.abs(r).max() < 1e-3
