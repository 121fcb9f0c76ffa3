"""Reference-held pin of the optimizer's OUTPUT conventions (pose signs, FOV <-> focal length in pixels, image-row order
of paramMap, pixel-edge NDC): the reference's own consumer code -- VideoDataset.update_poses (loaders/video_dataset.py:153-217)
and utils/geometry.py:62-138 -- was run on the drop-in's results and its outputs are committed
(tests/golden/reference_py/reprojection_golden.npz, minted by make_reprojection_golden.py next to it).

CPU: the numpy restatement of those reference lines (tests/reference_reprojection.py) reproduces the committed reference outputs;
with /root/reference mounted, the live reference functions map the synthetic generator's ground truth onto the flow targets.
GPU: a fresh drop-in run reproduces the dumped state and, through the pinned restatement, reprojects every static constraint
onto `pixel + flow` and the rendered depth up to one global scale.
"""
import importlib
import importlib.util
import os
import sys

import numpy as np
import pytest

from robust_cvd_amd import synth
from tests import baseline_configs as bc
from tests import margins
from tests import reference_reprojection as rr

# The case has an exact solution (zero flow noise; the depth error is a per-frame scale times a field on the optimizer's own
# 4 x 3 grid); what remains is the pull of the (reduced) regularisers, the stopping tolerance and the half-pixel between the
# corner-aligned sampling of paramMap (reference lib/DepthMapTransform.cpp:428-449) and the pixel-edge NDC of the constraints.
# Measured at mint time (fixture: `reprojection_error_px`): max 0.053 px, mean 0.0036 px.  The same check with paramMap's rows in
# the wrong order: max 0.56 px, mean 0.025 px (`flipped_rows_error_px_max`); a sign / axis / FOV error costs pixels.
# (3 x the measured values; the flipped-rows state still fails both by a factor of 3.3 / 2.2)
REPROJ_TOL_PX, REPROJ_MEAN_TOL_PX = 0.17, 0.011


def _golden():
    if not os.path.exists(rr.GOLDEN):
        pytest.skip("reprojection golden not minted")
    return dict(np.load(rr.GOLDEN))


def test_numpy_restatement_reproduces_the_reference_outputs():
    g = _golden()
    video = rr.make_case()
    assert bc.input_digest(video).encode() == g["input_sha256"].tobytes(), "the seeded case drifted from the minted one"
    ext, intr = rr.numpy_update_poses(g)
    np.testing.assert_allclose(ext, g["ref_extrinsics"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(intr, g["ref_intrinsics"], rtol=1e-6)
    # update_poses keeps paramMap as the float32 `scales` tensor, (H, W), row 0 = image top
    np.testing.assert_array_equal(g["param_map"].astype(np.float32), g["ref_scales"])
    assert float(g["ref_warp_abs_max"]) == 0.0  # identity spatial transform
    rp = rr.numpy_reproject(ext, intr, g["frame_a"], g["frame_b"], g["source_pixel"], g["source_depth_scaled"])
    assert np.abs(rp - g["ref_reprojected"]).max() < 2e-3  # (float32 arithmetic in both; pixels)
    # the reference's own functions put the optimizer's end state onto the flow targets
    err = np.linalg.norm(g["ref_reprojected"] - g["target_pixel"], axis=1)
    assert err.max() < REPROJ_TOL_PX and err.mean() < REPROJ_MEAN_TOL_PX, (err.max(), err.mean())
    assert float(g["flipped_rows_error_px_max"]) > 4 * err.max()  # (the check can tell the row order of paramMap)
    np.testing.assert_allclose(err, g["reprojection_error_px"], atol=1e-6)


def _mint_module():
    spec = importlib.util.spec_from_file_location(
        "make_reprojection_golden", os.path.join(os.path.dirname(rr.GOLDEN), "make_reprojection_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference checkout not mounted")
def test_live_reference_functions_on_the_generators_ground_truth():
    """The synthetic generator's TRUE cameras / scales, handed to the live reference code as if they were the optimizer's
    getters, reproject every constraint onto its flow target: generator, restatement and reference agree on every convention."""
    mk = _mint_module()
    v = rr.make_case()
    F, H, W = v.num_frames, v.height, v.width
    R = synth.rodrigues(v.true_w)
    out = dict(right=R[:, :, 0], up=R[:, :, 1], backward=R[:, :, 2], position=v.true_t,
               hfov=np.full(F, 2 * np.arctan(v.true_fy * v.aspect)), vfov=np.full(F, 2 * np.arctan(v.true_fy)),
               param_map=np.stack([v.frame_scale[f] * rr.grid_field(v.true_theta[f], W, H) for f in range(F)]),
               params=v.frame_scale[:, None],
               warp=np.zeros((F, H, W, 2), np.float32), source_depth=v.depth, width=np.int32(W), height=np.int32(H),
               depth_desc=np.frombuffer(b"Grid(Scale, Linear, 4, 3, 1)", np.uint8))
    ref = mk.reference_outputs(out, v)
    err = np.linalg.norm(ref["ref_reprojected"] - ref["target_pixel"], axis=1)
    assert err.max() < 1e-3, err.max()
    ext, intr = rr.numpy_update_poses(out)
    np.testing.assert_array_equal(ext, ref["ref_extrinsics"])
    np.testing.assert_allclose(intr, ref["ref_intrinsics"], rtol=1e-6)
    fa, fb, pix, _tgt, depth = rr.constraint_samples(v, out)
    assert np.abs(rr.numpy_reproject(ext, intr, fa, fb, pix, depth) - ref["ref_reprojected"]).max() < 1e-3


@pytest.mark.gpu
def test_drop_in_end_state_reprojects_through_the_reference_conventions(tmp_path):
    from robust_cvd_amd import build as _b
    d = os.path.dirname(_b.build_lib_python())
    if d not in sys.path:
        sys.path.insert(0, d)
    lib = importlib.import_module("lib_python")
    g = _golden()
    video = rr.make_case()
    out = rr.run_drop_in(lib, video, str(tmp_path / "video"))
    assert bytes(out["depth_desc"]) == bytes(g["depth_desc"])
    # the state the reference code was run on is the state this build produces
    perr, rerr = synth.relative_pose_error(out["position"], out["orientation"], g["position"], g["orientation"])
    # The fresh run against the MINTED state (another run, possibly of another build: a different workgroup size alone moves the end
    # state of this near-gauge case by 2e-4): the parity bar itself, 1e-3 -- what pins the conventions is the reprojection below.
    # (run-to-run spread of ONE build, eight runs: positions up to 5.7e-5, the scale-free depth scales up to 1.8e-4 --
    # profiles/r05_reproj_repeat.log; across two builds of this round: 2.1e-4)
    margins.below("position vs minted state", perr, 1e-3)
    margins.below("rotation vs minted state", rerr, 1e-3)
    margins.below("fov vs minted state", max(np.abs(out["vfov"] - g["vfov"]).max(), np.abs(out["hfov"] - g["hfov"]).max()), 1e-4)
    # right / up / backward are the columns of the pose's rotation matrix (what update_poses stacks into [R | t])
    Rm = synth.quat_to_matrix(out["orientation"])
    for k, name in enumerate(("right", "up", "backward")):
        np.testing.assert_allclose(out[name], Rm[:, :, k], atol=1e-6)
    # THE PIN, first: every static constraint reprojects onto its flow target through the reference's conventions
    ext, intr = rr.numpy_update_poses(out)
    fa, fb, pix, target, depth = rr.constraint_samples(video, out)
    err = np.linalg.norm(rr.numpy_reproject(ext, intr, fa, fb, pix, depth) - target, axis=1)
    margins.below("reprojection max px", err.max(), REPROJ_TOL_PX)
    margins.below("reprojection mean px", err.mean(), REPROJ_MEAN_TOL_PX)
    # NOT compared with the minted state: the depth scales.  scaleReg is 1e-6 in this case, so the overall scene scale (depth
    # scales and trajectory together) is a gauge direction of the cost: five runs of one build end 1.9e-3 apart in it
    # (profiles/r05_reproj_repeat.log) and two builds of this round -- the same arithmetic with another workgroup size -- 20 %
    # apart, while the similarity-aligned poses agree to 2e-4 and every constraint still reprojects to 0.053 px.  Round 4 compared
    # absolute scales here and its record went red for it; the scale-free statement is the last check of this test.
    gs_out, gs_g = np.median(out["params"]), np.median(g["params"])
    print(f"overall scale vs minted state: {gs_out / gs_g - 1.0:+.3e} (gauge, not asserted)")
    # depth x paramMap against the rendered depth, up to ONE global scale: only to a few percent per frame -- with nearly
    # parallel cameras and per-frame focal lengths the depth scale of a frame trades against its focal length (bas-relief
    # ambiguity; measured spread 2.4e-2 while every constraint reprojects to 0.05 px)
    spread, _worst = rr.depth_scale_spread(video, out)
    assert spread < 5e-2, spread
