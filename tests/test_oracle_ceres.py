"""The "true Ceres" column: when the oracle library is built against a real Ceres
(`make -C oracle CERES=1`, on a machine that has Ceres + Eigen -- this container has neither) the very same
residual blocks are solved by ceres::Solve with Ceres' own loss functions, corrector, trust-region loop and
SPARSE_NORMAL_CHOLESKY, and the oracle's restatement of those parts must reproduce it.  Skipped otherwise."""
import numpy as np
import pytest

from oracle import oracle as orc
from oracle.oracle import Oracle
from robust_cvd_amd import synth
from robust_cvd_amd.ctypes_types import OptParams, XformDesc


def test_has_ceres_reports_the_build():
    assert orc.has_ceres() in (True, False)
    if not orc.has_ceres():
        o = Oracle()
        o.set_linear_solver(2)
        v = synth.make_video(3, 64, 40, seed=1)
        synth.load_into(o, v)
        o.reset_depth_xforms(XformDesc.global_depth())
        o.reset_spatial_xforms(XformDesc.spatial())
        with pytest.raises(RuntimeError, match="without Ceres"):
            o.normalize_depth(OptParams.defaults())


@pytest.mark.skipif(not orc.has_ceres(), reason="oracle built without Ceres (make -C oracle CERES=1 where Ceres is installed)")
@pytest.mark.parametrize("variant", ["global_fixed", "ctf_6x4", "huber"])
def test_oracle_lm_reproduces_a_real_ceres_solve(variant):
    v = synth.make_video(12, 96, 56, seed=81)
    out = {}
    for kind in (0, 2):
        o = Oracle()
        o.set_linear_solver(kind)
        if variant == "huber":
            o.set_robust_loss(1)
        synth.load_into(o, v)
        p = OptParams.defaults()
        p.num_threads = 4
        if variant == "global_fixed":
            p.intr_opt, p.coarse_to_fine, p.num_steps = 0, 0, 1
        else:
            p.ctf_long, p.ctf_short = 6, 4
        o.reset_depth_xforms(XformDesc.global_depth())
        o.reset_spatial_xforms(XformDesc.spatial())
        o.normalize_depth(p)
        o.pose_optimization(p)
        out[kind] = (o.get_poses(), o.get_xform_params(), o.summary(), [r["cost"] for r in o.records()])
    a, b = out[0], out[2]
    assert a[2]["num_iterations"] == b[2]["num_iterations"]
    assert abs(a[2]["final_cost"] - b[2]["final_cost"]) <= 1e-9 * abs(b[2]["final_cost"])
    assert np.allclose(a[3], b[3], rtol=1e-8)   # cost after every iteration of the last level
    perr, rerr = synth.relative_pose_error(a[0]["position"], a[0]["orientation"], b[0]["position"], b[0]["orientation"])
    assert perr < 1e-5 and rerr < 1e-4
    assert np.abs(a[1] - b[1]).max() <= 1e-6 * np.abs(b[1]).max()
