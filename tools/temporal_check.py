#!/usr/bin/env python3
"""Development aid (GPU): the third preconditioner level (cvd_temporal.h) on a small problem -- its Galerkin matrix A_T as the
device assembled it (k_tl_diag / k_tl_edges / k_tl_reduce / k_tl_assemble) against P_T^T (J^T J + diag(lam)) P_T formed in numpy
from the matrix-free Hessian (cvd_evaluate's hfull) and an independently built P_T, and the inverse in use.
usage: temporal_check.py [frames] [step] [gx gy]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robust_cvd_amd import api, synth
from robust_cvd_amd.ctypes_types import OptParams, XformDesc


def axis_table(g, S):
    ratio = (g - 1) / (S - 1)
    b = np.zeros(g, dtype=int)
    h = np.zeros((g, 2), dtype=np.float32)
    for i in range(g):
        pos = i / ratio
        j0 = min(S - 2, max(0, int(np.floor(pos + 1e-12))))
        fr = min(1.0, max(0.0, pos - j0))
        b[i] = j0
        h[i] = (np.float32(1.0 - fr), np.float32(fr))
    return h, b


def prolongation(F, B, gx, gy, Sx, Sy, step):
    """P_T [F * B, S * nn]: column s * nn + a = (temporal hat a) x (coarse hat s), depth-grid rows only."""
    hx, bx = axis_table(gx, Sx)
    hy, by = axis_table(gy, Sy)
    S = Sx * Sy
    nn = (F - 1 + step - 1) // step + 1
    Hs = np.zeros((gx * gy, S))
    for vy in range(gy):
        for vx in range(gx):
            for i in range(2):
                for j in range(2):
                    Hs[vx + vy * gx, (bx[vx] + i) + (by[vy] + j) * Sx] += float(np.float32(hx[vx, i] * hy[vy, j]))
    P = np.zeros((F * B, S * nn))
    for f in range(F):
        for a in range(nn):
            w = max(0.0, 1.0 - abs(f - a * step) / step)
            if w > 0.0:
                P[f * B + 7:f * B + 7 + gx * gy, a::nn] += w * Hs   # columns s * nn + a for all s
    return P, S, nn


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 72
    step = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    gx, gy = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (5, 4)
    v = synth.make_video(F, 128, 72, seed=12, extra_offsets=6)
    s = api.Solver(0)
    synth.load_into(s, v)
    s.set_options(coarse_update_budget=0, temporal_level=2, temporal_step=step)
    s.reset_depth_xforms(XformDesc.global_depth())
    s.reset_spatial_xforms(XformDesc.spatial())
    p = OptParams.defaults()
    s.normalize_depth(p)
    s.grid_xform_split(XformDesc.grid_depth(gx, gy))
    reg = p.depth_deform_reg_final
    p.max_iterations = 3
    s.pose_optimization_step(p, reg, convert_poses=True)   # a state away from the initial one
    ev = s.evaluate(p, reg, None, want_hfull=True)
    p.max_iterations = 1
    s.pose_optimization_step(p, reg, convert_poses=False)  # ONE LM iteration: the level is built at the state just evaluated
    dbg = s.temporal_debug()
    assert dbg is not None, "the level was off"
    B = s.block_size()
    P, S, nn = prolongation(F, B, gx, gy, dbg["Sx"], dbg["Sy"], step)
    assert (S, nn, S * nn) == (dbg["S"], dbg["nn"], dbg["NT"]), (S, nn, dbg)
    H = ev["hfull"]
    ref = P.T @ (H + np.diag(dbg["lam"])) @ P
    A = dbg["a_t"].copy()
    A[np.diag_indices_from(A)] /= 1.0 + 1e-5
    scale = np.abs(ref).max()
    print(f"F {F} B {B} grid {gx}x{gy} -> {dbg['Sx']}x{dbg['Sy']} hats, {nn} nodes, NT {S * nn}; failed {dbg['failed']}")
    print("max |A_T - P^T (H + lam) P| / max |.| = %.3e   (diag part only: %.3e)" % (
        np.abs(A - ref).max() / scale, np.abs(np.diag(A) - np.diag(ref)).max() / scale))
    Hd = np.zeros_like(H)
    for f in range(F):
        Hd[f * B:(f + 1) * B, f * B:(f + 1) * B] = H[f * B:(f + 1) * B, f * B:(f + 1) * B]
    ref_d = P.T @ (Hd + np.diag(dbg["lam"])) @ P
    ref_p = P.T @ (H - Hd) @ P
    print("frame-diagonal part alone: max |ref| %.3e; pair part alone: max |ref| %.3e, max |(A_T - diag ref) - pair ref| %.3e" % (
        np.abs(ref_d).max(), np.abs(ref_p).max(), np.abs((A - ref_d) - ref_p).max()))
    i, j = np.unravel_index(np.argmax(np.abs(A - ref)), A.shape)
    print("largest deviation at (%d, %d): device %.9e reference %.9e (diag ref %.9e pair ref %.9e)" % (i, j, A[i, j], ref[i, j], ref_d[i, j], ref_p[i, j]))
    Ai = dbg["a_t_inverse"]
    print("max |A_T^-1 A_T - I| = %.3e, smallest eigenvalue of the inverse in use %.3e, cond(A_T) %.3e" % (
        np.abs(Ai @ dbg["a_t"] - np.eye(S * nn)).max(), np.linalg.eigvalsh(0.5 * (Ai + Ai.T))[0], np.linalg.cond(dbg["a_t"])))
    sm = s.summary()
    print("PCG iterations of that LM iteration:", sm["total_linear_iterations"])
    for lvl in (0, 2):
        s2 = api.Solver(0)
        synth.load_into(s2, v)
        s2.set_options(coarse_update_budget=0, temporal_level=lvl, temporal_step=step)
        s2.reset_depth_xforms(XformDesc.global_depth())
        s2.reset_spatial_xforms(XformDesc.spatial())
        q = OptParams.defaults()
        s2.normalize_depth(q)
        s2.grid_xform_split(XformDesc.grid_depth(gx, gy))
        s2.pose_optimization_step(q, reg, convert_poses=True)
        m = s2.summary()
        print(f"temporal_level {lvl}: LM {m['num_iterations']} PCG {m['total_linear_iterations']} final cost {m['final_cost']:.9e}")
        s2.close()


if __name__ == "__main__":
    main()
