#!/usr/bin/env python3
"""Development aid (CPU; oracle + scipy, see tools/pcg_lab.py): coarse-space candidates for the slow modes of an LM iteration --
per-frame spatial modes and spatial hats x TEMPORAL hat functions, joint Galerkin with the 8 modes per frame.
usage: pcg_lab_spaces.py <blocks.bin> <radius>      (results: profiles/r04_pcg_lab_recycling_and_temporal_level.log)"""
import sys, time, numpy as np, scipy.sparse as sp
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
import pcg_lab as L
path = sys.argv[1]; radius = float(sys.argv[2])
F, B, cost, g, I, J, blocks = L.load(path)
Aop = L.BlockOp(F, B, I, J, blocks)
hd = np.einsum("fii->fi", Aop.diag_blocks).ravel().copy()
lam = np.clip(hd, 1e-6, 1e32) / radius
b = -g
n = F * B
Dinv = np.linalg.inv(Aop.diag_blocks + np.einsum("fi,ij->fij", lam.reshape(F, B), np.eye(B)))
def bj(r): return np.einsum("fij,fj->fi", Dinv, r.reshape(F, B)).ravel()
Alam = Aop.A + sp.diags(lam)
gx, gy = 17, 10
G = gx * gy

def base_Z():
    rows, cols, vals = [], [], []
    for f in range(F):
        for i in range(7):
            rows.append(f * B + i); cols.append(f * 8 + i); vals.append(1.0)
        for v in range(G):
            rows.append(f * B + 7 + v); cols.append(f * 8 + 7); vals.append(1.0)
    return sp.csr_matrix((vals, (rows, cols)), shape=(n, F * 8))

def spatial(kind):
    tm = L.theta_modes(gx, gy, kind)[:, 1:]   # without the constant
    return tm

def per_frame_extra(tm):
    k = tm.shape[1]
    rows, cols, vals = [], [], []
    for f in range(F):
        for c in range(k):
            for v in range(G):
                rows.append(f * B + 7 + v); cols.append(f * k + c); vals.append(tm[v, c])
    return sp.csr_matrix((vals, (rows, cols)), shape=(n, F * k))

def temporal_extra(tm, step):
    k = tm.shape[1]
    nodes = np.arange(0, F + step - 1, step)
    nn = len(nodes)
    rows, cols, vals = [], [], []
    for f in range(F):
        for a, t in enumerate(nodes):
            w = max(0.0, 1.0 - abs(f - t) / step)
            if w == 0.0: continue
            for c in range(k):
                for v in range(G):
                    rows.append(f * B + 7 + v); cols.append(a * k + c); vals.append(w * tm[v, c])
    return sp.csr_matrix((vals, (rows, cols)), shape=(n, nn * k)), nn

def run(Z, label):
    t0 = time.time()
    AZ = Alam @ Z
    Ac = (Z.T @ AZ).toarray()
    Ac[np.diag_indices_from(Ac)] *= 1 + 1e-5
    Aci = np.linalg.inv(Ac)
    def M(r): return bj(r) + Z @ (Aci @ (Z.T @ r))
    _, it_own, h = L.pcg(Aop, lam, b, M, 1e-3)
    return it_own, h, time.time() - t0

Z0 = base_Z()
it0, h0, dt = run(Z0, "base")
m_inf = h0[-1]; delta = (m_inf - h0[it0 - 1]) / m_inf
print(f"base (8 modes / frame, n_c = {Z0.shape[1]}): {it0} iterations, model decrease {delta:.2e} short  [{dt:.0f} s]", flush=True)
def need(h):
    ok = np.flatnonzero((m_inf - h) / m_inf <= delta); return int(ok[0]) + 1 if len(ok) else -1
for label, Zx in (("quad per frame", lambda: per_frame_extra(spatial("quad"))),
                  ("quad x temporal hats every 8 frames", lambda: temporal_extra(spatial("quad"), 8)[0]),
                  ("quad x temporal hats every 16 frames", lambda: temporal_extra(spatial("quad"), 16)[0]),
                  ("grid4x3 x temporal hats every 8 frames", lambda: temporal_extra(L.theta_modes(gx, gy, "grid4x3"), 8)[0]),
                  ("grid6x4 x temporal hats every 8 frames", lambda: temporal_extra(L.theta_modes(gx, gy, "grid6x4"), 8)[0]),
                  ("grid6x4 x temporal hats every 32 frames", lambda: temporal_extra(L.theta_modes(gx, gy, "grid6x4"), 32)[0]),
                  ("all 170 vertices x temporal hats every 16 frames", lambda: temporal_extra(np.eye(G), 16)[0]),
                  ("all 170 vertices x temporal hats every 64 frames", lambda: temporal_extra(np.eye(G), 64)[0])):
    Ze = Zx()
    Z = sp.hstack([Z0, Ze]).tocsr()
    it_own, h, dt = run(Z, label)
    print(f"base + {label}: n_c = {Z.shape[1]} (+{Ze.shape[1]}): {need(h)} iterations to the same accuracy (own rule {it_own})  [{dt:.0f} s]", flush=True)
