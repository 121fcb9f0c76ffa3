#!/usr/bin/env python3
"""Development aid (CPU; oracle dump + scipy, see tools/pcg_lab.py): can the EXACT pose-graph level (8 modes per frame, 2400
unknowns at 300 frames, 1.6 ms per dense inverse on the device) be replaced by a TEMPORALLY COARSE one -- the frame's 8 modes x
temporal hat functions with a node every k frames -- now that the depth-grid patterns have a temporal level of their own?
usage: pcg_lab_temporal_pose.py <blocks.bin> <radius>"""
import os
import sys
import time
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pcg_lab as L

path = sys.argv[1]
radius = float(sys.argv[2])
F, B, cost, g, I, J, blocks = L.load(path)
Aop = L.BlockOp(F, B, I, J, blocks)
hd = np.einsum("fii->fi", Aop.diag_blocks).ravel().copy()
lam = np.clip(hd, 1e-6, 1e32) / radius
b = -g
n = F * B
Dinv = np.linalg.inv(Aop.diag_blocks + np.einsum("fi,ij->fij", lam.reshape(F, B), np.eye(B)))
Alam = Aop.A + sp.diags(lam)
gx, gy = 17, 10
G = gx * gy


def bj(r):
    return np.einsum("fij,fj->fi", Dinv, r.reshape(F, B)).ravel()


def frame_modes():
    """Z_f [B, 8]: identity on the 7 pose-like unknowns, the uniform depth-scale mode."""
    Zf = np.zeros((B, 8))
    Zf[:7, :7] = np.eye(7)
    Zf[7:, 7] = 1.0
    return Zf


def temporal(cols_per_frame, step):
    """[n, nn * m]: (per-frame columns, [B, m]) x temporal hats with a node every `step` frames (step 1: every frame its own)."""
    m = cols_per_frame.shape[1]
    nodes = np.arange(0, F + step - 1, step) if step > 1 else np.arange(F)
    rows, cols, vals = [], [], []
    rr, cc = np.nonzero(cols_per_frame)
    vv = cols_per_frame[rr, cc]
    for f in range(F):
        for a, t in enumerate(nodes):
            w = max(0.0, 1.0 - abs(f - t) / step)
            if w == 0.0:
                continue
            rows.append(f * B + rr)
            cols.append(a * m + cc)
            vals.append(w * vv)
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, len(nodes) * m))


def theta_hats(nx, ny):
    tm = L.theta_modes(gx, gy, f"grid{nx}x{ny}")
    out = np.zeros((B, tm.shape[1]))
    out[7:] = tm
    return out


def level(Z):
    Ac = (Z.T @ (Alam @ Z)).toarray()
    Ac[np.diag_indices_from(Ac)] *= 1 + 1e-5
    Aci = np.linalg.inv(Ac)
    return lambda r: Z @ (Aci @ (Z.T @ r))


REF = {}


def run(levels, label):
    """Iterations at the solver's own stopping rule, how far the model decrease is from its limit there, and the iterations that
    reach the accuracy the FIRST variant run (the round-3 preconditioner) has at its stop."""
    t0 = time.time()
    def M(r):
        z = bj(r)
        for lv in levels:
            z = z + lv(r)
        return z
    _, it, h = L.pcg(Aop, lam, b, M, 1e-3)
    m_inf = h[-1]
    short = (m_inf - h[it - 1]) / m_inf
    if not REF:
        REF["short"] = short
    same = int(np.argmax((m_inf - h) / m_inf <= REF["short"])) + 1
    print(f"{label}: {it} iterations at its own rule (model decrease {short:.2e} short), {same} to the reference accuracy  [{time.time() - t0:.0f} s]", flush=True)
    return it


Zf = frame_modes()
T_theta = level(temporal(theta_hats(9, 5), 32))
exact = level(temporal(Zf, 1))
print(f"F {F} B {B} radius {radius:g}")
run([exact, T_theta], "block-Jacobi + exact pose-graph level + depth-grid level 9x5 hats every 32 frames (the device now)")
for k in (16, 32, 4):
    Zk = temporal(Zf, k)
    run([level(Zk), T_theta], f"pose modes x hats every {k} frames ({Zk.shape[1]} unknowns) + depth-grid level")
Zp = np.zeros((B, 7)); Zp[:7, :7] = np.eye(7)
Zj = sp.hstack([temporal(Zp, 32), temporal(theta_hats(9, 5), 32)]).tocsr()
run([level(Zj)], f"JOINT, one matrix: (7 pose modes + 9x5 hats) x hats every 32 frames ({Zj.shape[1]} unknowns)")
