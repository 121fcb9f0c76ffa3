"""CPU pins of the oracle's block-sparse linear algebra (oracle/block_sparse.{h,cpp}) -- the exact solve that stands in for
the reference's SPARSE_NORMAL_CHOLESKY (reference lib/PoseOptimizer.cpp:956):
  * known answers against numpy on random block-sparse SPD matrices (ragged block sizes incl. empty blocks, fill-in,
    column scaling + extra diagonal as the LM loop uses them, the symmetric product);
  * the LM solve through the block-sparse factorisation equals the same solve through the dense Cholesky it replaced.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as orc
from oracle.oracle import Oracle
from robust_cvd_amd import synth
from robust_cvd_amd.ctypes_types import OptParams, XformDesc


def _solve(sizes, pairs, A, b, scale=None, extra=None, threads=4):
    lib = orc.load()
    sizes = np.ascontiguousarray(sizes, np.int32)
    pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    A = np.ascontiguousarray(A, np.float64)
    x = np.ascontiguousarray(b, np.float64).copy()
    y = np.zeros_like(x)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None
    sc = np.ascontiguousarray(scale, np.float64) if scale is not None else None
    ex = np.ascontiguousarray(extra, np.float64) if extra is not None else None
    rc = lib.cvdo_block_sparse_solve(C.c_int(len(sizes)), sizes.ctypes.data_as(C.POINTER(C.c_int)), C.c_int(len(pairs)),
                                     pairs.ctypes.data_as(C.POINTER(C.c_int)), dp(A), dp(sc), dp(ex), dp(x), dp(y),
                                     C.c_int(threads))
    return rc, x, y


def _random_block_spd(rng, sizes, pairs):
    off = np.concatenate([[0], np.cumsum(sizes)])
    n = off[-1]
    # J with one row group per pair / per block: A = J^T J + eps I has exactly the requested block structure
    rows = []
    for (i, j) in list(pairs) + [(k, k) for k in range(len(sizes))]:
        r = np.zeros((max(3, sizes[i] + sizes[j]), n))
        r[:, off[i]:off[i + 1]] = rng.normal(size=(r.shape[0], sizes[i]))
        r[:, off[j]:off[j + 1]] = rng.normal(size=(r.shape[0], sizes[j]))
        rows.append(r)
    J = np.concatenate(rows, 0)
    return J.T @ J + 1e-3 * np.eye(n)


@pytest.mark.parametrize("case", ["chain", "hierarchical", "ragged_with_empty", "dense_graph"])
def test_block_cholesky_known_answers(case):
    rng = np.random.default_rng({"chain": 1, "hierarchical": 2, "ragged_with_empty": 3, "dense_graph": 4}[case])
    if case == "chain":
        sizes = [5] * 12
        pairs = [(i, i + 1) for i in range(11)]
    elif case == "hierarchical":
        sizes = [23] * 40  # (23 = 7 + 16: the block of BASELINE configs[1]; exercises the ragged edges of the 6x16 micro-kernel)
        pairs = sorted({(min(a, b), max(a, b)) for a, b in synth.hierarchical_pairs(40)})
    elif case == "ragged_with_empty":
        sizes = [7, 0, 19, 1, 33, 0, 8, 17, 2, 40]
        pairs = [(0, 2), (2, 4), (4, 9), (3, 7), (7, 9), (0, 9), (6, 8), (2, 8), (1, 5)]
    else:
        sizes = [9] * 10
        pairs = [(i, j) for i in range(10) for j in range(i + 1, 10)]
    A = _random_block_spd(rng, sizes, [(i, j) for i, j in pairs if sizes[i] and sizes[j]])
    n = A.shape[0]
    b = rng.normal(size=n)
    rc, x, y = _solve(sizes, pairs, A, b)
    assert rc >= len(sizes)
    assert np.abs(y - A @ b).max() <= 1e-12 * np.abs(A @ b).max()
    assert np.abs(x - np.linalg.solve(A, b)).max() <= 1e-9 * np.abs(x).max()
    # the LM system: column scaling and the damping diagonal
    scale = 1.0 / (1.0 + np.sqrt(np.diag(A)))
    extra = rng.uniform(0.1, 1.0, size=n)
    rc, x2, _ = _solve(sizes, pairs, A, b, scale, extra)
    M = A * scale[:, None] * scale[None, :] + np.diag(extra)
    assert np.abs(x2 - np.linalg.solve(M, b)).max() <= 1e-9 * np.abs(x2).max()


def test_block_cholesky_reports_indefinite_matrices():
    sizes = [3, 3]
    A = np.eye(6)
    A[4, 4] = -1.0
    rc, _, _ = _solve(sizes, [(0, 1)], A, np.ones(6))
    assert rc == -1


def test_sparse_and_dense_lm_solves_agree():
    """Full coarse-to-fine solve of a small video: block-sparse Cholesky (default) vs the dense Cholesky it replaced."""
    v = synth.make_video(10, 96, 56, seed=5)
    out = {}
    for kind in (0, 1):
        o = Oracle()
        o.set_linear_solver(kind)
        synth.load_into(o, v)
        p = OptParams.defaults()
        p.num_threads = 4
        p.ctf_long, p.ctf_short = 6, 4
        o.reset_depth_xforms(XformDesc.global_depth())
        o.reset_spatial_xforms(XformDesc.spatial())
        o.normalize_depth(p)
        o.pose_optimization(p)
        out[kind] = (o.get_pose_params(), o.get_xform_params(), o.summary(), [r["cost"] for r in o.records()])
    assert out[0][2]["num_iterations"] == out[1][2]["num_iterations"]
    assert np.allclose(out[0][3], out[1][3], rtol=1e-10, atol=0)
    assert np.abs(out[0][0] - out[1][0]).max() < 1e-8
    assert np.abs(out[0][1] - out[1][1]).max() < 1e-8
