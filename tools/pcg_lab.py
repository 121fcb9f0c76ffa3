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
    o.reset_depth_xforms(XformDesc.grid_depth(gx, gy))
    o.reset_spatial_xforms(XformDesc.spatial())
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


def pcg(Aop, lam, b, Minv, eta=1e-3, maxit=400, record=None):
    x = np.zeros_like(b)
    r = b.copy()
    z = Minv(r)
    p = z.copy()
    rz = r @ z
    rz0 = rz
    its = 0
    while its < maxit:
        q = Aop(p) + lam * p
        if record is not None:
            record.append((z.copy(), ))
        alpha = rz / (p @ q)
        x += alpha * p
        r -= alpha * q
        z = Minv(r)
        rz_new = r @ z
        its += 1
        if rz_new <= eta * eta * rz0:
            break
        p = z + (rz_new / rz) * p
        rz = rz_new
    return x, its


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


def main_run(path, radius=1e4, eta=1e-3):
    F, B, cost, g, I, J, blocks = load(path)
    print(f"F {F} B {B} blocks {len(I)} cost {cost:.6f} |g| {np.abs(g).max():.3e}")
    Aop = BlockOp(F, B, I, J, blocks)
    hd = np.einsum("fii->fi", Aop.diag_blocks).ravel().copy()
    lam = np.clip(hd, 1e-6, 1e32) / radius
    b = -g
    Dinv = np.linalg.inv(Aop.diag_blocks + np.einsum("fi,ij->fij", lam.reshape(F, B), np.eye(B)))

    def bj(r):
        return np.einsum("fij,fj->fi", Dinv, r.reshape(F, B)).ravel()

    t0 = time.time()
    _, it = pcg(Aop, lam, b, bj, eta)
    print(f"block-Jacobi only: {it} iterations ({time.time() - t0:.1f} s)")
    gx, gy = {177: (17, 10), 91: (12, 7), 31: (6, 4), 199: (16, 12)}[B]

    def coarse_setup(kind, shift=1e-5):
        tm = theta_modes(gx, gy, kind)
        Zf, m = build_Z(F, B, tm)
        # A_c = Z^T (A + lam) Z, block (f, g) = Zf^T A_fg Zf
        Ac = np.zeros((F * m, F * m))
        for k in range(len(I)):
            blk = Zf.T @ blocks[k] @ Zf
            i, j = I[k], J[k]
            Ac[i * m:(i + 1) * m, j * m:(j + 1) * m] += blk
            if i != j:
                Ac[j * m:(j + 1) * m, i * m:(i + 1) * m] += blk.T
        for f in range(F):
            Ac[f * m:(f + 1) * m, f * m:(f + 1) * m] += Zf.T @ (lam.reshape(F, B)[f][:, None] * Zf)
        Ac[np.diag_indices_from(Ac)] *= (1.0 + shift)
        Aci = np.linalg.inv(Ac)
        return Zf, m, Aci

    for kind in sys.argv[3:] or ["const", "tilt", "bilinear", "quad", "grid3x2", "grid3x3", "grid4x3"]:
        t0 = time.time()
        Zf, m, Aci = coarse_setup(kind)

        def additive(r):
            rc = (r.reshape(F, B) @ Zf).ravel()
            c = (Aci @ rc).reshape(F, m)
            return bj(r) + (c @ Zf.T).ravel()

        _, it_add = pcg(Aop, lam, b, additive, eta)

        # multiplicative (symmetric): coarse correction, block-Jacobi on the new residual, coarse correction again
        def coarse(r):
            rc = (r.reshape(F, B) @ Zf).ravel()
            return ((Aci @ rc).reshape(F, m) @ Zf.T).ravel()

        def mult(r):
            z1 = coarse(r)
            r1 = r - (Aop(z1) + lam * z1)
            z2 = z1 + bj(r1)
            r2 = r - (Aop(z2) + lam * z2)
            return z2 + coarse(r2)

        _, it_mul = pcg(Aop, lam, b, mult, eta, maxit=200)
        print(f"coarse '{kind}' ({m} modes / frame, n_c = {F * m}): additive {it_add} its, multiplicative {it_mul} its "
              f"(3 products each)  [{time.time() - t0:.1f} s]")


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
    else:
        main_run(sys.argv[2])
