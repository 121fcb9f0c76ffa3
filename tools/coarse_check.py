"""Development aid (GPU): numerical check of the coarse preconditioner level against dense algebra."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robust_cvd_amd import api, synth
from robust_cvd_amd.ctypes_types import OptParams, XformDesc

F = int(sys.argv[1]) if len(sys.argv) > 1 else 24
gx, gy = 6, 4
v = synth.make_video(F, 192, 112, seed=7)


def run(coarse):
    s = api.Solver(0)
    synth.load_into(s, v)
    s.set_options(coarse_level=coarse)
    s.reset_depth_xforms(XformDesc.global_depth())
    s.reset_spatial_xforms(XformDesc.spatial())
    p = OptParams.defaults()
    s.normalize_depth(p)
    s.pose_optimization_step(p, 0.0)
    a = s.summary()
    s.grid_xform_split(XformDesc.grid_depth(gx, gy))
    p.max_iterations = 3
    s.pose_optimization_step(p, p.depth_deform_reg_initial)
    b = s.summary()
    return s, p, a, b


def run_global(coarse):
    s = api.Solver(0)
    synth.load_into(s, v)
    s.set_options(coarse_level=coarse, verbose=int(os.environ.get("CVD_VERBOSE", "1")))
    s.reset_depth_xforms(XformDesc.global_depth())
    s.reset_spatial_xforms(XformDesc.spatial())
    p = OptParams.defaults()
    s.normalize_depth(p)
    p.max_iterations = 4
    s.pose_optimization_step(p, 0.0)
    return s


sg = run_global(1)
dg = sg.coarse_debug()
Ag, Aig = dg["a_c"], dg["a_c_inverse"]
print("GLOBAL: n", Ag.shape[0], "failed", dg["failed"], "eig min %.3e" % np.linalg.eigvalsh(Ag)[0],
      "|Ainv A - I|", np.abs(Aig @ Ag - np.eye(Ag.shape[0])).max())
pg = OptParams.defaults()
Hg = sg.evaluate(pg, 0.0, want_hfull=True)["hfull"]
ng = Ag.shape[0]
offm = np.ones((ng, ng), bool)
for f in range(F):
    offm[f * 8:(f + 1) * 8, f * 8:(f + 1) * 8] = False
print("GLOBAL off-diagonal A_c vs H: rel %.3e ; diagonal blocks (A_c - H) should be diagonal lam: offdiag-in-block max %.3e, lam range %.3e..%.3e" % (
    np.abs((Ag - Hg)[offm]).max() / np.abs(Hg[offm]).max(),
    max(np.abs((Ag - Hg)[f * 8:(f + 1) * 8, f * 8:(f + 1) * 8] - np.diag(np.diag((Ag - Hg)[f * 8:(f + 1) * 8, f * 8:(f + 1) * 8]))).max() for f in range(F)),
    np.diag(Ag - Hg).min(), np.diag(Ag - Hg).max()))
run_global(0)
s1, p, a1, b1 = run(1)
dbg = s1.coarse_debug()
s0, _, a0, b0 = run(0)
print("global level: coarse its %d lin %d cost %.9g | plain its %d lin %d cost %.9g" % (
    a1["num_iterations"], a1["total_linear_iterations"], a1["final_cost"],
    a0["num_iterations"], a0["total_linear_iterations"], a0["final_cost"]))
print("grid level  : coarse its %d lin %d cost %.9g | plain its %d lin %d cost %.9g" % (
    b1["num_iterations"], b1["total_linear_iterations"], b1["final_cost"],
    b0["num_iterations"], b0["total_linear_iterations"], b0["final_cost"]))
A, Ai = dbg["a_c"], dbg["a_c_inverse"]
n = A.shape[0]
print("n", n, "failed", dbg["failed"], "sym A", np.abs(A - A.T).max(), "sym Ainv", np.abs(Ai - Ai.T).max() / np.abs(Ai).max())
ev = np.linalg.eigvalsh(A)
print("A_c eig min %.3e max %.3e" % (ev[0], ev[-1]))
print("|Ainv A - I| max", np.abs(Ai @ A - np.eye(n)).max(), " vs numpy inverse rel", np.abs(Ai - np.linalg.inv(A)).max() / np.abs(Ai).max())
# off-diagonal blocks against Z^T (J^T J) Z from the matrix-free product at the same state (state of s1 after the solve
# differs from the linearisation point of the last LM iteration, so evaluate on a fresh comparison: only structure/scale)
ev1 = s1.evaluate(p, p.depth_deform_reg_initial, want_hfull=True)
H = ev1["hfull"]
B = s1.block_size()
Z = np.zeros((F * B, n))
for f in range(F):
    for i in range(7):
        Z[f * B + i, f * 8 + i] = 1
    Z[f * B + 7:(f + 1) * B, f * 8 + 7] = 1
AcH = Z.T @ H @ Z
mask = np.ones((n, n), bool)
for f in range(F):
    mask[f * 8:(f + 1) * 8, f * 8:(f + 1) * 8] = False
print("off-diagonal blocks vs Z^T H Z at the final state: rel diff %.3e (differs by one LM step)" % (
    np.abs((A - AcH)[mask]).max() / np.abs(AcH[mask]).max()))
