#!/usr/bin/env python3
"""Development aid (CPU only, uses the ORACLE as evaluator -- never part of the product path): preconditioner experiments for
the PCG of the LM step on the real benchmark problem.

    python tools/pcg_lab.py dump config2_4k /tmp/lab/blocks.bin      # oracle: block-sparse J^T J + gradient at a state
    python tools/pcg_lab.py run /tmp/lab/blocks.bin                  # PCG iteration counts of the preconditioner variants

The device solver's preconditioner is restated in numpy / scipy (block-Jacobi + additive coarse level on Z = 8 modes per frame,
dense coarse inverse with the 1e-5 diagonal shift) so that its iteration count can be compared with the candidates' at the
solver's stopping rule sqrt(r^T M^-1 r) <= eta * initial, eta = 1e-3.
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dump(name, path, lib_path=None):
    from oracle import oracle as orc
    if lib_path:
        orc._LIB_PATH = lib_path
        orc.build = lambda force=False: lib_path
    from robust_cvd_amd.ctypes_types import XformDesc
    from robust_cvd_amd import synth
    from tests import baseline_configs as bc
    video = bc.make_video(name)
    ref = bc.load_solution(name)
    o = orc.Oracle()
    p = bc.params_for(name, threads=8)
    synth.load_into(o, video, p.focal_long)
    gx, gy = int(ref["grid_size"][0]), int(ref["grid_size"][1])
    o.reset_depth_xforms(XformDesc.global_depth())
    o.reset_spatial_xforms(XformDesc.spatial())
    if os.environ.get("LAB_STATE", "pipeline") == "pipeline":
        # the state the final coarse-to-fine level STARTS from (what bench.py's timed iterations see): the earlier levels solved
        import bench
        o.normalize_depth(p)
        grids = bench.ctf_schedule(p, video.aspect)
        first = True
        for grid in [None] + grids[:-1]:
            if grid is not None:
                o.grid_xform_split(XformDesc.grid_depth(*grid))
            t1 = time.time()
            o.pose_optimization_step(p, p.depth_deform_reg_final, convert_poses=first)
            print(f"level {grid}: {time.time() - t1:.1f} s, cost {o.summary()['final_cost']:.6f}", flush=True)
            first = False
        o.grid_xform_split(XformDesc.grid_depth(*grids[-1]))
        pose7 = o.get_pose_params().copy()
    else:
        o.reset_depth_xforms(XformDesc.grid_depth(gx, gy))
        rng = np.random.default_rng(5)
        pose7 = ref["pose7"] + rng.normal(0.0, 1e-3, size=ref["pose7"].shape)
        theta = ref["depth_params"] * (1.0 + rng.normal(0.0, 1e-2, size=ref["depth_params"].shape))
        o.set_xform_params(theta)
    t0 = time.time()
    pp = np.ascontiguousarray(pose7, np.float64)
    rc = o._fn("dump_blocks")(o._h, C.byref(p), C.c_double(p.depth_deform_reg_final), pp.ctypes.data_as(C.POINTER(C.c_double)),
                              path.encode())
    o._check(rc)
    print(f"dumped {path} in {time.time() - t0:.1f} s, grid {gx}x{gy}")


def load(path):
    with open(path, "rb") as f:
        F, B = np.fromfile(f, np.int32, 2)
        nb = int(np.fromfile(f, np.int64, 1)[0])
        cost = float(np.fromfile(f, np.float64, 1)[0])
        g = np.fromfile(f, np.float64, F * B)
        I = np.empty(nb, np.int32)
        J = np.empty(nb, np.int32)
        blocks = np.empty((nb, B, B), np.float64)
        for k in range(nb):
            ij = np.fromfile(f, np.int32, 2)
            I[k], J[k] = ij
            blocks[k] = np.fromfile(f, np.float64, B * B).reshape(B, B)
    return int(F), int(B), cost, g, I, J, blocks


class BlockOp:
    """Symmetric block-sparse matrix (lower blocks given) as a matvec."""

    def __init__(self, F, B, I, J, blocks):
        import scipy.sparse as sp
        self.F, self.B = F, B
        off = I != J
        rows = np.concatenate([I, J[off]])
        cols = np.concatenate([J, I[off]])
        data = np.concatenate([blocks, blocks[off].transpose(0, 2, 1)])
        order = np.lexsort((cols, rows))
        rows, cols, data = rows[order], cols[order], data[order]
        indptr = np.zeros(F + 1, np.int64)
        np.add.at(indptr, rows + 1, 1)
        indptr = np.cumsum(indptr)
        self.A = sp.bsr_matrix((data, cols, indptr), shape=(F * B, F * B))
        self.diag_blocks = np.zeros((F, B, B))
        d = I == J
        self.diag_blocks[I[d]] = blocks[d]

    def __call__(self, x):
        return self.A @ x


def pcg(Aop, lam, b, Minv, eta=1e-3, maxit=400):
    """Returns x, iterations at the solver's stopping rule, and the model decrease after every iteration (CG lowers
    phi(x) = x^T A x / 2 - b^T x by alpha r^T z / 2 per iteration: the energy error is 2 (m_inf - m_k))."""
    x = np.zeros_like(b)
    r = b.copy()
    z = Minv(r)
    p = z.copy()
    rz = r @ z
    rz0 = rz
    its, stop_it, m, hist = 0, None, 0.0, []
    while its < maxit:
        q = Aop(p) + lam * p
        alpha = rz / (p @ q)
        x += alpha * p
        r -= alpha * q
        m += 0.5 * alpha * rz
        hist.append(m)
        z = Minv(r)
        rz_new = r @ z
        its += 1
        if stop_it is None and rz_new <= eta * eta * rz0:
            stop_it = its
        if rz_new <= 1e-22 * rz0:
            break
        p = z + (rz_new / rz) * p
        rz = rz_new
    return x, (stop_it or its), np.array(hist)


def theta_modes(gx, gy, kind):
    """Per-frame coarse modes on the depth grid (vertex row-major, x fastest): columns of a [G, m] matrix."""
    xs = np.linspace(-1, 1, gx)
    ys = np.linspace(-1, 1, gy)
    X, Y = np.meshgrid(xs, ys)
    X, Y = X.ravel(), Y.ravel()
    one = np.ones_like(X)
    sets = {
        "const": [one],
        "tilt": [one, X, Y],
        "bilinear": [one, X, Y, X * Y],
        "quad": [one, X, Y, X * Y, X * X - 1 / 3, Y * Y - 1 / 3],
    }
    if kind in sets:
        return np.stack(sets[kind], 1)
    if kind.startswith("grid"):  # gridNxM: bilinear hat functions of a coarse NxM grid
        n, m = map(int, kind[4:].split("x"))
        def hats(t, k):
            c = np.linspace(-1, 1, k)
            h = 2.0 / (k - 1)
            return np.clip(1 - np.abs(t[:, None] - c[None]) / h, 0, None)
        hx, hy = hats(X, n), hats(Y, m)
        return (hx[:, :, None] * hy[:, None, :]).reshape(len(X), n * m)
    raise ValueError(kind)


def build_Z(F, B, tm):
    """Z [F*B, F*m]: per frame [I7 0; 0 theta modes]."""
    G = B - 7
    m = 7 + tm.shape[1]
    Zf = np.zeros((B, m))
    Zf[:7, :7] = np.eye(7)
    Zf[7:, 7:] = tm
    return Zf, m


def main_run(path, kinds, radii=(1e4, 3e4, 9e4, 2.7e5), eta=1e-3):
    F, B, cost, g, I, J, blocks = load(path)
    print(f"F {F} B {B} blocks {len(I)} cost {cost:.6f} |g| {np.abs(g).max():.3e}", flush=True)
    Aop = BlockOp(F, B, I, J, blocks)
    hd = np.einsum("fii->fi", Aop.diag_blocks).ravel().copy()
    b = -g
    gx, gy = {177: (17, 10), 91: (12, 7), 31: (6, 4), 199: (16, 12)}[B]

    # coarse Galerkin blocks Zf^T A_fg Zf per kind (lam-independent part)
    def galerkin(kind):
        tm = theta_modes(gx, gy, kind)
        Zf, m = build_Z(F, B, tm)
        Ac = np.zeros((F * m, F * m))
        for k in range(len(I)):
            blk = Zf.T @ blocks[k] @ Zf
            i, j = I[k], J[k]
            Ac[i * m:(i + 1) * m, j * m:(j + 1) * m] += blk
            if i != j:
                Ac[j * m:(j + 1) * m, i * m:(i + 1) * m] += blk.T
        return Zf, m, Ac

    gal = {k: galerkin(k) for k in kinds}
    for radius in radii:
        lam = np.clip(hd, 1e-6, 1e32) / radius
        Dinv = np.linalg.inv(Aop.diag_blocks + np.einsum("fi,ij->fij", lam.reshape(F, B), np.eye(B)))

        def bj(r):
            return np.einsum("fij,fj->fi", Dinv, r.reshape(F, B)).ravel()

        def two_level(kind, shift=1e-5):
            Zf, m, Ac0 = gal[kind]
            Ac = Ac0.copy()
            for f in range(F):
                Ac[f * m:(f + 1) * m, f * m:(f + 1) * m] += Zf.T @ (lam.reshape(F, B)[f][:, None] * Zf)
            Ac[np.diag_indices_from(Ac)] *= (1.0 + shift)
            Aci = np.linalg.inv(Ac)

            def additive(r):
                rc = (r.reshape(F, B) @ Zf).ravel()
                return bj(r) + ((Aci @ rc).reshape(F, m) @ Zf.T).ravel()
            return additive, m

        # the solver's preconditioner at its stopping rule defines the accuracy the others must reach
        base, _ = two_level("const")
        _, it_base, hist = pcg(Aop, lam, b, base, eta)
        m_inf = hist[-1]
        delta = (m_inf - hist[it_base - 1]) / m_inf
        print(f"radius {radius:.1e}: baseline (8 modes, additive) stops after {it_base} iterations with the model decrease "
              f"{delta:.2e} short of exact ({len(hist)} iterations to 1e-11)", flush=True)

        def needed(h):
            short = (m_inf - h) / m_inf
            ok = np.flatnonzero(short <= delta)
            return int(ok[0]) + 1 if len(ok) else -1

        _, _, h = pcg(Aop, lam, b, bj, eta)
        print(f"    block-Jacobi only: {needed(h)} iterations to the same accuracy", flush=True)
        for kind in kinds:
            if kind == "const":
                continue
            t0 = time.time()
            M, m = two_level(kind)
            _, it_own, h = pcg(Aop, lam, b, M, eta)
            print(f"    coarse '{kind}' ({m} modes / frame, n_c = {F * m}): {needed(h)} iterations to the same accuracy "
                  f"(own stopping rule: {it_own})  [{time.time() - t0:.1f} s]", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
    else:
        main_run(sys.argv[2], sys.argv[3:] or ["const", "tilt", "quad", "grid3x3", "grid4x3", "grid6x4"])
