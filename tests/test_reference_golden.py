"""The route to REAL-reference golden vectors: `video.dat` files written by the reference's own Ceres build
(tools/make_reference_golden.py, on a machine that has one) are read back with robust_cvd_amd.dataset_io.read_video_dat and
compared with this repository's solves of the same seeded inputs.

* CPU: the reader is pinned against this repository's own writer (lib_python's DepthVideo.save, same format 13).
* When tests/golden/reference_ceres/<config>/video.dat exists: the ORACLE's end state must match it (CPU: this is what
  turns "parity unpinned" into a pinned oracle) and so must the HIP path (GPU).  Without such a file the two tests skip --
  none can be produced here: no Ceres, no network."""
import importlib
import json
import os
import sys

import numpy as np
import pytest

from robust_cvd_amd import build as _b
from robust_cvd_amd import dataset_io, synth
from tests import baseline_configs as bc

REF_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_ceres")


@pytest.fixture(scope="module")
def lib():
    d = os.path.dirname(_b.build_lib_python())
    if d not in sys.path:
        sys.path.insert(0, d)
    return importlib.import_module("lib_python")


def test_video_dat_reader_round_trips_our_writer(lib, tmp_path):
    from tests.drop_in_caller import build_pose_optimizer
    v = synth.make_video(6, 96, 56, seed=71)
    base = dataset_io.write_dataset(str(tmp_path / "video"), v)
    dv, _ = build_pose_optimizer(lib, base, "midas2", list(range(v.num_frames)), None)
    ds = dv.depthStream(0)
    g = lib.XformDescriptor()
    g.type, g.depthType, g.valueXform = lib.XformType.Depth, lib.DepthXformType.Grid, lib.ValueXformType.Scale
    g.gridSize = [4, 3, 1]
    ds.resetDepthXforms(g)
    rng = np.random.default_rng(1)
    want = []
    for f in range(v.num_frames):
        fr = ds.frame(f)
        th = list(rng.uniform(0.5, 2.0, 12))
        fr.depthXform().setParams(th)
        e = fr.extrinsics
        e.position = [float(x) for x in rng.normal(size=3)]
        q = rng.normal(size=4)
        e.orientation.setCoeffs([float(x) for x in q / np.linalg.norm(q)])
        fr.extrinsics = e
        want.append((np.asarray(e.position, np.float32), np.asarray(e.orientation.coeffs(), np.float32), fr.intrinsics.vFov, th))
    dv.save()
    vd = dataset_io.read_video_dat(os.path.join(base, "video.dat"))
    assert vd["version"] == 13 and len(vd["pts"]) == v.num_frames and (vd["width"], vd["height"]) == (v.width, v.height)
    assert [c["name"] for c in vd["color_streams"]][:2] == ["full", "down"]
    st = vd["depth_streams"][0]
    assert st["name"] == "depth_midas2" and st["depth_desc"] == "Grid(Scale, Linear, 4, 3, 1)" and st["spatial_desc"] == "Identity"
    pos, quat, vfov, theta = dataset_io.poses_from_video_dat(vd)
    for f, (p, q, fov, th) in enumerate(want):
        assert np.array_equal(pos[f], p) and np.array_equal(quat[f], q) and vfov[f] == np.float32(fov)
        assert np.array_equal(theta[f], np.asarray(th))


def _reference_cases():
    if not os.path.isdir(REF_DIR):
        return []
    return sorted(d for d in os.listdir(REF_DIR) if os.path.exists(os.path.join(REF_DIR, d, "video.dat")))


def _load_reference(name):
    meta = json.load(open(os.path.join(REF_DIR, name, "meta.json")))
    video = bc.make_video(meta["config"])
    if bc.input_digest(video) != meta["input_sha256"]:
        pytest.skip("the synthetic inputs regenerated here differ from the ones the reference file was minted on")
    return meta, video, dataset_io.poses_from_video_dat(dataset_io.read_video_dat(os.path.join(REF_DIR, name, "video.dat")))


@pytest.mark.parametrize("name", _reference_cases() or ["<none committed>"])
def test_oracle_matches_the_real_reference(name):
    if name.startswith("<"):
        pytest.skip("no tests/golden/reference_ceres/*/video.dat committed (mint with tools/make_reference_golden.py)")
    from oracle.oracle import Oracle
    meta, video, (pos, quat, vfov, theta) = _load_reference(name)
    sol = bc.run(Oracle(), meta["config"], video)
    perr, rerr = synth.relative_pose_error(sol["position"], sol["orientation"], pos, quat)
    assert perr <= 1e-3 and rerr <= 1e-3, (perr, rerr)
    assert np.abs(sol["vfov"] - vfov).max() <= 1e-4
    assert np.abs(sol["depth_params"] - theta).max() <= 1e-3 * np.abs(theta).max()


@pytest.mark.gpu
@pytest.mark.parametrize("name", _reference_cases() or ["<none committed>"])
def test_hip_matches_the_real_reference(name):
    if name.startswith("<"):
        pytest.skip("no tests/golden/reference_ceres/*/video.dat committed (mint with tools/make_reference_golden.py)")
    from robust_cvd_amd import api
    meta, video, (pos, quat, vfov, theta) = _load_reference(name)
    sol = bc.run(api.Solver(0), meta["config"], video)
    perr, rerr = synth.relative_pose_error(sol["position"], sol["orientation"], pos, quat)
    assert perr <= 1e-3 and rerr <= 1e-3, (perr, rerr)
    assert np.abs(sol["vfov"] - vfov).max() <= 1e-4
    assert np.abs(sol["depth_params"] - theta).max() <= 1e-3 * np.abs(theta).max()
