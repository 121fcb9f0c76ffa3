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
    np.savez(path + ".state.npz", pose7=pose7, theta=o.get_xform_params())   # (for `next`: the state one LM step further)
    t0 = time.time()
    pp = np.ascontiguousarray(pose7, np.float64)
    rc = o._fn("dump_blocks")(o._h, C.byref(p), C.c_double(p.depth_deform_reg_final), pp.ctypes.data_as(C.POINTER(C.c_double)),
                              path.encode())
    o._check(rc)
    print(f"dumped {path} in {time.time() - t0:.1f} s, grid {gx}x{gy}")


def dump_next(name, path, path_next, lib_path=None, radius=1e4):
    """The problem ONE LM step further: the (tightly solved) first step of the dumped problem at `radius` is applied to its state
    and the oracle dumps blocks + gradient there -- what the next LM iteration's linear system looks like (recycling experiments)."""
    from oracle import oracle as orc
    if lib_path:
        orc._LIB_PATH = lib_path
        orc.build = lambda force=False: lib_path
    from robust_cvd_amd.ctypes_types import XformDesc
    from robust_cvd_amd import synth
    from tests import baseline_configs as bc
    F, B, cost, g, I, J, blocks = load(path)
    st = np.load(path + ".state.npz")
    Aop = BlockOp(F, B, I, J, blocks)
    hd = np.einsum("fii->fi", Aop.diag_blocks).ravel().copy()
    lam = np.clip(hd, 1e-6, 1e32) / radius
    Dinv = np.linalg.inv(Aop.diag_blocks + np.einsum("fi,ij->fij", lam.reshape(F, B), np.eye(B)))
    dx, _, _ = pcg(Aop, lam, -g, lambda r: np.einsum("fij,fj->fi", Dinv, r.reshape(F, B)).ravel(), 1e-3, maxit=600)
    dx = dx.reshape(F, B)
    video = bc.make_video(name)
    o = orc.Oracle()
    p = bc.params_for(name, threads=8)
    synth.load_into(o, video, p.focal_long)
    gx, gy = {177: (17, 10), 91: (12, 7), 31: (6, 4), 199: (16, 12)}[B]
    o.reset_depth_xforms(XformDesc.grid_depth(gx, gy))
    o.reset_spatial_xforms(XformDesc.spatial())
    pose7 = st["pose7"] + dx[:, :7]
    o.set_xform_params(st["theta"] + dx[:, 7:])
    np.savez(path_next + ".state.npz", pose7=pose7, theta=o.get_xform_params())
    pp = np.ascontiguousarray(pose7, np.float64)
    o._check(o._fn("dump_blocks")(o._h, C.byref(p), C.c_double(p.depth_deform_reg_final), pp.ctypes.data_as(C.POINTER(C.c_double)),
                                  path_next.encode()))
    print(f"dumped {path_next} (|dx| max {np.abs(dx).max():.3e})")


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


# ---- deflation with Ritz vectors (recycling): python tools/pcg_lab.py deflate blocks.bin [k ...] ---------------------------------
def pcg_lanczos(Aop, lam, b, Minv, its):
    """`its` PCG iterations recording the M-orthonormal Lanczos basis u_j = z_j / sqrt(r_j^T z_j) and the tridiagonal T = U^T A U."""
    n = len(b)
    x = np.zeros(n)
    r = b.copy()
    z = Minv(r)
    p = z.copy()
    rz = r @ z
    U, alphas, betas = [], [], []
    for _ in range(its):
        U.append(z / np.sqrt(rz))
        q = Aop(p) + lam * p
        alpha = rz / (p @ q)
        x += alpha * p
        r -= alpha * q
        z = Minv(r)
        rz_new = r @ z
        beta = rz_new / rz
        alphas.append(alpha)
        betas.append(beta)
        p = z + beta * p
        rz = rz_new
    k = len(alphas)
    T = np.zeros((k, k))
    for j in range(k):
        T[j, j] = 1.0 / alphas[j] + (betas[j - 1] / alphas[j - 1] if j > 0 else 0.0)
        if j + 1 < k:
            T[j, j + 1] = T[j + 1, j] = -np.sqrt(betas[j]) / alphas[j]
    return np.array(U).T, T


def deflated_pcg(Aop, lam, b, Minv, W, eta_hist_target=None, maxit=400):
    """Deflated PCG (Saad, Yeung, Erhel, Guyomarc'h 2000) with the deflation space W; returns the model-decrease history
    (including the decrease of the initial guess x0 = W E^-1 W^T b as entry 0)."""
    AW = np.stack([Aop(W[:, j]) + lam * W[:, j] for j in range(W.shape[1])], 1)
    E = W.T @ AW
    Ei = np.linalg.inv(0.5 * (E + E.T))
    x = W @ (Ei @ (W.T @ b))
    r = b - (Aop(x) + lam * x)
    m0 = b @ x - 0.5 * x @ (Aop(x) + lam * x)
    z = Minv(r)
    p = z - W @ (Ei @ (AW.T @ z))
    rz = r @ z
    rz0 = rz
    m, hist = m0, [m0]
    for _ in range(maxit):
        q = Aop(p) + lam * p
        alpha = rz / (p @ q)
        x += alpha * p
        r -= alpha * q
        m += 0.5 * alpha * rz
        hist.append(m)
        z = Minv(r)
        rz_new = r @ z
        if rz_new <= 1e-22 * rz0:
            break
        p = z - W @ (Ei @ (AW.T @ z)) + (rz_new / rz) * p
        rz = rz_new
    return np.array(hist)


def main_deflate(path, ks, eta=1e-3):
    F, B, cost, g, I, J, blocks = load(path)
    Aop = BlockOp(F, B, I, J, blocks)
    hd = np.einsum("fii->fi", Aop.diag_blocks).ravel().copy()
    b = -g
    gx, gy = {177: (17, 10), 91: (12, 7), 31: (6, 4), 199: (16, 12)}[B]
    tm = theta_modes(gx, gy, "const")
    Zf, m = build_Z(F, B, tm)
    Ac0 = np.zeros((F * m, F * m))
    for k in range(len(I)):
        blk = Zf.T @ blocks[k] @ Zf
        i, j = I[k], J[k]
        Ac0[i * m:(i + 1) * m, j * m:(j + 1) * m] += blk
        if i != j:
            Ac0[j * m:(j + 1) * m, i * m:(i + 1) * m] += blk.T

    def precond(radius):
        lam = np.clip(hd, 1e-6, 1e32) / radius
        Dinv = np.linalg.inv(Aop.diag_blocks + np.einsum("fi,ij->fij", lam.reshape(F, B), np.eye(B)))
        Ac = Ac0.copy()
        for f in range(F):
            Ac[f * m:(f + 1) * m, f * m:(f + 1) * m] += Zf.T @ (lam.reshape(F, B)[f][:, None] * Zf)
        Ac[np.diag_indices_from(Ac)] *= 1.0 + 1e-5
        Aci = np.linalg.inv(Ac)

        def M(r):
            rc = (r.reshape(F, B) @ Zf).ravel()
            return np.einsum("fij,fj->fi", Dinv, r.reshape(F, B)).ravel() + ((Aci @ rc).reshape(F, m) @ Zf.T).ravel()
        return lam, M

    # Ritz vectors from the solve of the FIRST LM iteration (radius 1e4, as many iterations as the device runs)
    lam1, M1 = precond(1e4)
    _, it1, hist1 = pcg(Aop, lam1, b, M1, eta)
    U, T = pcg_lanczos(Aop, lam1, b, M1, it1)
    theta, Y = np.linalg.eigh(T)
    print(f"first solve: {it1} iterations; Ritz values of M^-1 A: smallest {theta[:6].round(4)}, largest {theta[-3:].round(3)}", flush=True)
    for radius in (1e4, 3e4):
        lam, M = precond(radius)
        _, it_base, hist = pcg(Aop, lam, b, M, eta)
        m_inf = hist[-1]
        delta = (m_inf - hist[it_base - 1]) / m_inf
        print(f"radius {radius:.0e}: plain PCG {it_base} iterations (model decrease {delta:.2e} short)", flush=True)
        for k in ks:
            W = U @ Y[:, :k]                       # the k Ritz vectors of the smallest Ritz values
            h = deflated_pcg(Aop, lam, b, M, W)
            short = (m_inf - h) / m_inf
            ok = np.flatnonzero(short <= delta)
            print(f"    deflated with {k:2d} Ritz vectors of the first solve: {int(ok[0]) if len(ok) else -1} iterations to the same "
                  f"accuracy (+ {k} products for A W)", flush=True)
        for k in ks[:2]:
            W = U @ Y[:, -k:]                      # ... of the LARGEST Ritz values, for comparison
            h = deflated_pcg(Aop, lam, b, M, W)
            short = (m_inf - h) / m_inf
            ok = np.flatnonzero(short <= delta)
            print(f"    deflated with the {k:2d} LARGEST Ritz vectors: {int(ok[0]) if len(ok) else -1} iterations", flush=True)


# ---- spectral (additive low-rank) recycling across LM iterations: python tools/pcg_lab.py recycle blocks_s1.bin blocks_s2.bin ----
def main_recycle(path1, path2, ks=(4, 8, 16), eta=1e-3):
    """Ritz pairs (theta_i, w_i) of M^-1 A from the Lanczos data of the PCG solve of LM iteration 1 (state 1, radius 1e4) are reused in
    LM iteration 2 (state 2 = state 1 + step, radius 3e4, its own block-Jacobi + coarse level) as an additive low-rank term
        M_2s^-1 = M_2^-1 + W diag(1 / theta - 1) W^T        (W M-orthonormal: w_i^T A w_i = theta_i)
    -- positive semi-definite whatever happened to A in between, no product with the new A needed."""
    def setup(path, radius):
        F, B, cost, g, I, J, blocks = load(path)
        Aop = BlockOp(F, B, I, J, blocks)
        hd = np.einsum("fii->fi", Aop.diag_blocks).ravel().copy()
        gx, gy = {177: (17, 10), 91: (12, 7), 31: (6, 4), 199: (16, 12)}[B]
        Zf, m = build_Z(F, B, theta_modes(gx, gy, "const"))
        lam = np.clip(hd, 1e-6, 1e32) / radius
        Dinv = np.linalg.inv(Aop.diag_blocks + np.einsum("fi,ij->fij", lam.reshape(F, B), np.eye(B)))
        Ac = np.zeros((F * m, F * m))
        for k in range(len(I)):
            blk = Zf.T @ blocks[k] @ Zf
            i, j = I[k], J[k]
            Ac[i * m:(i + 1) * m, j * m:(j + 1) * m] += blk
            if i != j:
                Ac[j * m:(j + 1) * m, i * m:(i + 1) * m] += blk.T
        for f in range(F):
            Ac[f * m:(f + 1) * m, f * m:(f + 1) * m] += Zf.T @ (lam.reshape(F, B)[f][:, None] * Zf)
        Ac[np.diag_indices_from(Ac)] *= 1.0 + 1e-5
        Aci = np.linalg.inv(Ac)

        def M(r):
            rc = (r.reshape(F, B) @ Zf).ravel()
            return np.einsum("fij,fj->fi", Dinv, r.reshape(F, B)).ravel() + ((Aci @ rc).reshape(F, m) @ Zf.T).ravel()
        return Aop, lam, -g, M, cost

    A1, lam1, b1, M1, c1 = setup(path1, 1e4)
    _, it1, _ = pcg(A1, lam1, b1, M1, eta)
    U, T = pcg_lanczos(A1, lam1, b1, M1, it1)
    theta, Y = np.linalg.eigh(T)
    print(f"LM iteration 1 (cost {c1:.6f}): {it1} PCG iterations; smallest Ritz values {theta[:8].round(4)}", flush=True)
    A2, lam2, b2, M2, c2 = setup(path2, 3e4)
    _, it2, hist = pcg(A2, lam2, b2, M2, eta)
    m_inf = hist[-1]
    delta = (m_inf - hist[it2 - 1]) / m_inf
    print(f"LM iteration 2 (cost {c2:.6f}, radius 3e4): plain PCG {it2} iterations (model decrease {delta:.2e} short)", flush=True)
    for k in ks:
        W = U @ Y[:, :k]
        gain = 1.0 / theta[:k] - 1.0

        def Ms(r, W=W, gain=gain):
            return M2(r) + W @ (gain * (W.T @ r))
        _, it_own, h = pcg(A2, lam2, b2, Ms, eta)
        short = (m_inf - h) / m_inf
        ok = np.flatnonzero(short <= delta)
        print(f"    + {k:2d} recycled Ritz pairs (additive spectral term): {int(ok[0]) + 1 if len(ok) else -1} iterations to the same "
              f"accuracy (own stopping rule: {it_own})", flush=True)
        h2 = deflated_pcg(A2, lam2, b2, M2, W)
        short = (m_inf - h2) / m_inf
        ok = np.flatnonzero(short <= delta)
        print(f"      exact deflation with the same {k} vectors (+ {k} products for A W): {int(ok[0]) if len(ok) else -1} iterations", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
    elif sys.argv[1] == "next":
        dump_next(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else None)
    elif sys.argv[1] == "recycle":
        main_recycle(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "deflate":
        main_deflate(sys.argv[2], [int(a) for a in sys.argv[3:]] or [4, 8, 16, 32])
    else:
        main_run(sys.argv[2], sys.argv[3:] or ["const", "tilt", "quad", "grid3x3", "grid4x3", "grid6x4"])
