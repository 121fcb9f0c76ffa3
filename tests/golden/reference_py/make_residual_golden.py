#!/usr/bin/env python3
"""Mint tests/golden/reference_py/residual_golden.npz: the reference's OWN torch statement of the flow-constraint residual
(utils/geometry.py:62-166, loss/consistency_loss.py:27-199) evaluated, in float64, on the seeded NON-converged states of
tests/reference_residuals.py.  Runs in the build container only (needs /root/reference and torch; no GPU):

    python tests/golden/reference_py/make_residual_golden.py

Per case `<name>/...`: frames [n, 2], pose [F, 7] (the state), pixel_diff [n, 2] (pixels), disparity_diff [n],
loss_reproj / loss_disp [n] (what ConsistencyLoss.geometry_consistency_loss returns per constraint), cost_gradient_fd [F, 7] (central
differences of the cost formed from the reference terms along every pose parameter and focal length), fd_rows (indices of the
constraints whose derivative rows are kept), fd [len(fd_rows), 3, 14] (central differences of the torch functions along
[pose_a(6) | pose_b(6) | vfocal_a | vfocal_b]), fd_depth [len(fd_rows), 3, 2] (along the deformed depths D_a, D_b) and
cost_gradient_theta_fd [F, nD] (central differences of the reference cost along every depth-transform parameter).  The oracle-side inputs (NDC, deformed depths) are recomputed by the test
from the same seed; the input digest of every case's video is stored so that a drifted generator is noticed.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)

from tests import baseline_configs as bc  # noqa: E402
from tests import reference_residuals as rres  # noqa: E402

FD_EVERY = 8


def main():
    out = {}
    for name in rres.CASES:
        ref = rres.reference_outputs(name)
        v, _o, _p, _pose = rres.make_state(name)
        rows = np.arange(0, len(ref["pixel_diff"]), FD_EVERY)
        out[name + "/input_sha256"] = np.frombuffer(bc.input_digest(v).encode(), np.uint8)
        for k in ("frames", "pose", "pixel_diff", "disparity_diff", "loss_reproj", "loss_disp"):
            out[f"{name}/{k}"] = ref[k]
        out[name + "/cost_gradient_fd"] = rres.reference_cost_gradient(name)
        out[name + "/cost_gradient_theta_fd"] = rres.reference_cost_gradient_theta(name)
        out[name + "/fd_rows"] = rows.astype(np.int32)
        out[name + "/fd"] = ref["fd"][rows]
        out[name + "/fd_depth"] = ref["fd_depth"][rows]
        print(f"{name}: {len(ref['pixel_diff'])} constraints, |pixel diff| up to {np.abs(ref['pixel_diff']).max():.2f} px, "
              f"|disparity diff| up to {np.abs(ref['disparity_diff']).max():.4f}")
    np.savez_compressed(rres.GOLDEN, **out)
    print("wrote", rres.GOLDEN, os.path.getsize(rres.GOLDEN), "bytes")


if __name__ == "__main__":
    main()
