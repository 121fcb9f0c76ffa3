#!/usr/bin/env python3
"""Development aid (CPU; oracle + scipy, see tools/pcg_lab.py): structure of the smallest Ritz vectors of the preconditioned operator
(energy split pose / focal / depth grid, share in the per-frame mean, temporal smoothness, spatial pattern).
usage: pcg_lab_ritz_look.py <blocks.bin> <radius> <lanczos steps>"""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
import pcg_lab as L
path = sys.argv[1]; radius = float(sys.argv[2]); nit = int(sys.argv[3])
F, B, cost, g, I, J, blocks = L.load(path)
Aop = L.BlockOp(F, B, I, J, blocks)
hd = np.einsum("fii->fi", Aop.diag_blocks).ravel().copy()
Zf, m = L.build_Z(F, B, L.theta_modes(17, 10, "const"))
lam = np.clip(hd, 1e-6, 1e32) / radius
Dinv = np.linalg.inv(Aop.diag_blocks + np.einsum("fi,ij->fij", lam.reshape(F, B), np.eye(B)))
Ac = np.zeros((F * m, F * m))
for k in range(len(I)):
    blk = Zf.T @ blocks[k] @ Zf; i, j = I[k], J[k]
    Ac[i*m:(i+1)*m, j*m:(j+1)*m] += blk
    if i != j: Ac[j*m:(j+1)*m, i*m:(i+1)*m] += blk.T
for f in range(F): Ac[f*m:(f+1)*m, f*m:(f+1)*m] += Zf.T @ (lam.reshape(F, B)[f][:, None] * Zf)
Ac[np.diag_indices_from(Ac)] *= 1 + 1e-5
Aci = np.linalg.inv(Ac)
def M(r):
    rc = (r.reshape(F, B) @ Zf).ravel()
    return np.einsum("fij,fj->fi", Dinv, r.reshape(F, B)).ravel() + ((Aci @ rc).reshape(F, m) @ Zf.T).ravel()
U, T = L.pcg_lanczos(Aop, lam, -g, M, nit)
th, Y = np.linalg.eigh(T)
print("ritz", th[:8].round(4))
for i in range(4):
    w = (U @ Y[:, i]).reshape(F, B)
    e = w ** 2
    tot = e.sum()
    pose, foc, grid = e[:, :6].sum() / tot, e[:, 6].sum() / tot, e[:, 7:].sum() / tot
    gm = w[:, 7:]
    mean_part = (gm.mean(1) ** 2 * gm.shape[1]).sum() / max(e[:, 7:].sum(), 1e-300)   # share of the grid energy in the per-frame mean
    # temporal smoothness: energy of frame-to-frame differences relative to energy
    dt = ((w[1:] - w[:-1]) ** 2).sum() / tot
    # spatial pattern of the grid part (avg over frames of |value|), 10 rows x 17 cols
    pat = np.sqrt((gm ** 2).mean(0)).reshape(10, 17)
    fe = e.sum(1) / tot
    print(f"vec {i}: theta {th[i]:.4f} energy pose {pose:.3f} focal {foc:.3f} grid {grid:.3f}; grid energy in per-frame mean {mean_part:.3f}; temporal diff/energy {dt:.3f}; top frames {np.argsort(-fe)[:6]} share {np.sort(fe)[::-1][:6].round(3)}")
    print("   spatial rms pattern rows(bottom..top) x cols, normalised:\n", (pat / pat.max()).round(2))
