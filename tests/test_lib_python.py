"""Drop-in `lib_python` module (robust_cvd_amd/csrc/lib_python.cpp): the names / semantics the reference's
Python callers rely on, the on-disk formats, and -- on the GPU -- the full optimize_poses() sequence."""
import importlib
import os
import struct
import sys
import types

import numpy as np
import pytest

from robust_cvd_amd import build as _b
from robust_cvd_amd import dataset_io, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    d = os.path.dirname(_b.build_lib_python())
    if d not in sys.path:
        sys.path.insert(0, d)
    return importlib.import_module("lib_python")


@pytest.fixture()
def dataset(tmp_path):
    v = synth.make_video(10, 96, 56, seed=51)
    return v, dataset_io.write_dataset(str(tmp_path / "video"), v)


def test_names_imported_by_the_reference_callers_exist(lib):
    # pose_optimization.py:9-23, params.py:14, process.py:25-30, loaders/video_dataset.py:16
    for n in ("DepthVideo", "DepthVideoImporter", "DepthVideoPoseOptimizer", "DepthVideoProcessor", "DepthXformType",
              "FlowConstraintsCollection", "FlowConstraintsParams", "IntrinsicsOptimization", "SmoothLossType",
              "SpatialXformType", "StaticLossType", "ValueXformType", "XformType", "FrameRange", "initLib", "logToStdout"):
        assert hasattr(lib, n), n
    p = lib.DepthVideoPoseOptimizer.Params()   # defaults of reference lib/PoseOptimizer.h:55-103 (params.py mirrors them)
    assert (p.maxIterations, p.numThreads, p.numSteps, p.robustness) == (1000, 12, 4, 0.5)
    assert p.staticLossType == lib.StaticLossType.ReproDisparity and p.intrOpt == lib.IntrinsicsOptimization.PerFrame
    assert (p.ctfLong, p.ctfShort, p.dsoLong, p.dsoShort, p.scaleRegGridSize) == (17, 10, 4, 3, 10)
    assert p.focalLong == 0.3461538376301239 and p.depthDeformRegFinal == 0.1 and p.coarseToFine


def test_frame_range(lib):
    r = lib.FrameRange()
    r.fromString("1,3,5-7")
    assert r.toString() == "1,3,5-7" and r.count() == 5 and r.firstFrame() == 1 and r.lastFrame() == 7
    assert r.inRange(6) and not r.inRange(4) and not r.isConsecutive()
    r2 = lib.FrameRange()
    assert r2.isEmpty()
    with pytest.raises(RuntimeError):
        r2.firstFrame()
    r2.resolve(4)
    assert r2.toString() == "0-3"
    r3 = lib.FrameRange()
    r3.fromString("2-9")
    r3.resolve(5, True)
    assert r3.toString() == "2-4"
    with pytest.raises(RuntimeError):
        r.resolve(5)


def test_xform_descriptor_strings(lib):
    d = lib.XformDescriptor()
    assert d.str() == "Identity()"
    d.depthType = lib.DepthXformType.Global
    d.valueXform = lib.ValueXformType.Scale
    assert d.str() == "Global(Scale)"
    d.depthType = lib.DepthXformType.Grid
    d.gridSize = [17, 10, 1]
    assert d.str() == "Grid(Scale, Linear, 17, 10, 1)"
    e = lib.XformDescriptor()
    e.parse("Grid(ScaleShift, Cubic, 4, 3, 1)")
    assert e.cubicInterpolation and list(e.gridSize) == [4, 3, 1] and e.valueXform == lib.ValueXformType.ScaleShift
    e.parse("BicubicGrid(Scale, 5, 6)")          # legacy 2-D form, reference lib/DepthMapTransform.cpp:197-207
    assert e.str() == "Grid(Scale, Cubic, 5, 6, 1)"
    s = lib.XformDescriptor()
    s.reset(lib.XformType.Spatial)
    assert s.str() == "Identity"
    s.parse("BicubicGrid(3, 4)")
    assert s.spatialType == lib.SpatialXformType.BicubicGrid and s.str() == "BicubicGrid(3, 4)"


def test_nested_params_are_references_and_assignment_copies(lib):
    """pose_optimization.py:190-216 relies on both behaviours (SURVEY.md 8b 'ownership')."""
    opt = lib.DepthVideoPoseOptimizer.Params()
    params = lib.DepthVideoProcessor.Params()
    params.poseOptimizer = opt
    params.poseOptimizer.frameRange.fromString("0-3")     # mutates in place through the getter
    assert params.poseOptimizer.frameRange.toString() == "0-3"
    params.poseOptimizer.fixPoses = True
    assert opt.fixPoses is False                           # the setter copied
    params.depthXformDesc.depthType = lib.DepthXformType.Global
    assert params.depthXformDesc.depthType == lib.DepthXformType.Global


def test_members_write_through_and_clone(lib):
    """The reference binds Eigen members (numpy views): element assignment on gridSize / depthMinMax / position writes
    through (lib/PythonBindings.cpp:181, 237-238); Xform.clone (:246) copies descriptor and parameters."""
    d = lib.XformDescriptor()
    d.type, d.depthType, d.valueXform = lib.XformType.Depth, lib.DepthXformType.Grid, lib.ValueXformType.Scale
    d.gridSize = [4, 3, 1]
    d.gridSize[0] = 6
    d.gridSize[1] += 2
    assert list(d.gridSize) == [6, 5, 1] and d.str() == "Grid(Scale, Linear, 6, 5, 1)"
    d.depthMinMax[1] = 7.5
    assert list(d.depthMinMax) == [0.0, 7.5]
    view = d.gridSize
    del d
    view[2] = 1  # the view keeps its parent alive
    e = lib.Extrinsics()
    e.position[2] = 3.0
    e.position = [1.0, 2.0, e.position[2]]
    assert list(e.position) == [1.0, 2.0, 3.0]


def test_xform_clone(lib, dataset):
    v, base = dataset
    from tests.drop_in_caller import build_pose_optimizer
    dv, _ = build_pose_optimizer(lib, base, "midas2", list(range(v.num_frames)), None)
    ds = dv.depthStream(0)
    g = lib.XformDescriptor()
    g.type, g.depthType, g.valueXform = lib.XformType.Depth, lib.DepthXformType.Grid, lib.ValueXformType.Scale
    g.gridSize = [3, 2, 1]
    ds.resetDepthXforms(g)
    x = ds.frame(2).depthXform()
    x.setParams([1.0, 2.0, 3.0, 4.0, 5.0, 6.0])
    c = x.clone()
    assert isinstance(c, lib.DepthXform) and c.desc().str() == x.desc().str() and c.params() == x.params()
    x.setParams([9.0] * 6)
    assert c.params() == [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]   # a copy, owned by Python
    ds.frame(3).depthXform().copyFrom(c)
    assert ds.frame(3).depthXform().params() == c.params()
    s = ds.frame(2).spatialXform().clone()
    assert isinstance(s, lib.SpatialXform) and s.numParams() == 0


def test_prune_static_flag(lib, dataset):
    """pruneStaticFlag (reference lib/FlowConstraints.cpp:662-748) against a plain numpy restatement: disks of radius
    `distance` around the end points of the non-static constraints, then every constraint touching a stamped pixel."""
    v, base = dataset
    from tests.drop_in_caller import build_pose_optimizer
    dv, fc = build_pose_optimizer(lib, base, "midas2", list(range(v.num_frames)), None)
    w, h, F = v.width, v.height, v.num_frames
    rng = np.random.default_rng(3)
    keys = [tuple(int(x) for x in p) for p in v.pairs]
    flags = {}
    for a, b in keys:
        n = len(fc.staticFlags(a, b))
        f = (rng.uniform(size=n) > 0.03).astype(np.uint8)  # a few dynamic constraints per pair
        fc.setStaticFlags(a, b, list(f))
        flags[(a, b)] = f
    dist = 4
    masks = np.zeros((F, h, w), bool)
    yy, xx = np.mgrid[0:h, 0:w]
    locs = {k: np.asarray(fc.pairLocations(*k)) for k in keys}
    for (a, b) in keys:
        for i in np.nonzero(flags[(a, b)] == 0)[0]:
            for fr, (lx, ly) in ((a, locs[(a, b)][i, 0:2]), (b, locs[(a, b)][i, 2:4])):
                x, y = int(np.float32(lx) * np.float32(w)), int(np.float32(ly) * np.float32(w))
                masks[fr] |= (xx - x) ** 2 + (yy - y) ** 2 <= dist * dist
    fc.pruneStaticFlag(dist)
    changed = 0
    for (a, b) in keys:
        l = locs[(a, b)]
        x0, y0 = (l[:, 0] * np.float32(w)).astype(int), (l[:, 1] * np.float32(w)).astype(int)
        x1, y1 = (l[:, 2] * np.float32(w)).astype(int), (l[:, 3] * np.float32(w)).astype(int)
        want = flags[(a, b)].astype(bool) & ~(masks[a][y0, x0] | masks[b][y1, x1])
        got = np.asarray(fc.staticFlags(a, b), bool)
        assert np.array_equal(got, want), (a, b)
        changed += int((flags[(a, b)].astype(bool) & ~want).sum())
    assert changed > 0


def test_import_streams_constraints_and_video_dat(lib, dataset):
    v, base = dataset
    from tests.drop_in_caller import build_pose_optimizer
    dv, fc = build_pose_optimizer(lib, base, "midas2", list(range(v.num_frames)), None)
    assert dv.numFrames() == 10 and dv.width() == 96 and dv.height() == 56
    assert abs(dv.aspect() - 96 / 56) < 1e-6 and dv.numDepthStreams() == 1 and dv.hasColorStream("down")
    ds = dv.depthStream(dv.numDepthStreams() - 1)
    assert ds.name() == "depth_midas2" and ds.depthXformDesc().str() == "Identity()"
    src = ds.frame(3).sourceDepth()                       # disparity file -> depth, invalid -> 0
    assert src.shape == (56, 96) and ds.width() == 96
    np.testing.assert_allclose(src, v.depth[3], rtol=2e-7)
    f = ds.frame(0)
    assert f.extrinsics.right() == [1.0, 0.0, 0.0] and f.extrinsics.backward() == [0.0, 0.0, 1.0]
    assert f.intrinsics.vFov > 0 and f.intrinsics.hFov > 0   # resolveMissingFov
    assert fc.numPairs() == len(v.pairs) and fc.numConstraints() == v.num_constraints
    # video.dat layout: magic, version 13, dp format 3, N, pts ... trailing magic
    raw = open(os.path.join(base, "video.dat"), "rb").read()
    magic, ver, dpf, n = struct.unpack_from("<IIIi", raw, 0)
    assert (magic, ver, dpf, n) == (0xDEADBEEF, 13, 3, 10)
    assert struct.unpack_from("<I", raw, len(raw) - 4)[0] == 0xDEADBEEF
    dur, w, h, asp, iasp, magic2 = struct.unpack_from("<fiiffI", raw, len(raw) - 24)
    assert (w, h, magic2) == (96, 56, 0xDEADBEEF) and abs(asp - 96 / 56) < 1e-6 and abs(iasp - 56 / 96) < 1e-6
    assert b"Identity()" in raw and b"depth_midas2" in raw
    # flow_constraints.dat round trip is byte-identical
    before = open(os.path.join(base, "flow_constraints.dat"), "rb").read()
    fc.save()
    assert open(os.path.join(base, "flow_constraints.dat"), "rb").read() == before


def test_clip_max_depth_op(lib, dataset):
    """Op.ClipMaxDepth, reference lib/Processor.cpp:592-617: depth <- min(depth, maxDepth) for the frames of the range."""
    v, base = dataset
    dv = lib.DepthVideo()
    lib.DepthVideoImporter.importVideo(dv, base, True)
    dv.createDepthStream("depth_midas2", "depth_midas2", [v.width, v.height])
    proc = lib.DepthVideoProcessor(dv)
    params = lib.DepthVideoProcessor.Params()
    params.frameRange.fromString("2-4")
    params.op = lib.DepthVideoProcessor.Op.ClipMaxDepth
    params.depthStream = 0
    params.maxDepth = float(np.median(v.depth[3]))
    before = dv.depthStream(0).frame(5).depth().copy()
    proc.process(params)
    d3 = dv.depthStream(0).frame(3).depth()
    assert d3.max() <= params.maxDepth and np.array_equal(d3, np.minimum(v.depth[3], np.float32(params.maxDepth)).astype(np.float32)) \
        or np.allclose(d3, np.minimum(v.depth[3], params.maxDepth), rtol=3e-7)
    assert np.array_equal(dv.depthStream(0).frame(5).depth(), before)


def test_missing_inputs_raise_runtime_error(lib, tmp_path):
    dv = lib.DepthVideo()
    with pytest.raises(RuntimeError, match="frame file"):
        lib.DepthVideoImporter.importVideo(dv, str(tmp_path), False)
    v = synth.make_video(4, 48, 28, seed=52, spacing=8)
    base = dataset_io.write_dataset(str(tmp_path / "v"), v)
    os.remove(os.path.join(base, "flow_list.json"))
    lib.DepthVideoImporter.importVideo(dv, base, False)
    fcp = lib.FlowConstraintsParams()
    fcp.frameRange.resolve(dv.numFrames(), True)
    with pytest.raises(RuntimeError, match="Flow list file does not exist"):
        lib.FlowConstraintsCollection(dv, fcp)


def test_param_map_and_warp_match_the_oracle_gathers(lib, dataset):
    """DepthXform.paramMap / SpatialXform.warp (corner-aligned sampling, reference lib/DepthMapTransform.cpp:950-994,
    428-449) against the oracle's gather at the same locations."""
    from oracle import oracle as orc
    from robust_cvd_amd.ctypes_types import SpatialXformType, XformDesc
    v, base = dataset
    dv = lib.DepthVideo()
    lib.DepthVideoImporter.importVideo(dv, base, False)
    dv.createDepthStream("depth_midas2", "depth_midas2", [-1, -1])
    ds = dv.depthStream(0)
    d = lib.XformDescriptor()
    d.depthType = lib.DepthXformType.Grid
    d.valueXform = lib.ValueXformType.Scale
    d.gridSize = [5, 4, 1]
    ds.resetDepthXforms(d)
    rng = np.random.default_rng(0)
    theta = rng.uniform(0.5, 1.5, 20)
    ds.frame(2).depthXform().setParams(theta.tolist())
    pm = np.asarray(ds.frame(2).depthXform().paramMap(ds.frame(2)))
    assert pm.shape == (56, 96) and pm.dtype == np.float64
    od = XformDesc.grid_depth(5, 4)
    for (y, x) in ((0, 0), (55, 95), (17, 40), (30, 3)):
        lx = np.float32(-1) + np.float32(x) * (np.float32(2) / np.float32(95))
        ly = np.float32(1) - np.float32(y) * (np.float32(2) / np.float32(55))
        idx, w = orc.gather(od, 1.0, float(lx), float(ly))
        assert abs(pm[y, x] - float(theta[idx] @ w)) < 1e-12
    s = lib.XformDescriptor()
    s.reset(lib.XformType.Spatial)
    s.spatialType = lib.SpatialXformType.BicubicGrid
    s.gridSize = [4, 3, 0]
    ds.resetSpatialXforms(s)
    phi = rng.normal(0, 0.01, 24)
    ds.frame(1).spatialXform().setParams(phi.tolist())
    wmap = np.asarray(ds.frame(1).spatialXform().warp(56, 96))
    assert wmap.shape == (56, 96, 2) and wmap.dtype == np.float32
    osd = XformDesc.spatial(SpatialXformType.BicubicGrid, 4, 3)
    for (y, x) in ((0, 0), (55, 95), (20, 50)):
        lx = np.float32(-1) + np.float32(x) * (np.float32(2) / np.float32(95))
        ly = np.float32(1) - np.float32(y) * (np.float32(2) / np.float32(55))
        idx, w = orc.gather(osd, 1.0, float(lx), float(ly))
        exp = (phi.reshape(-1, 2)[idx] * w[:, None]).sum(0)
        np.testing.assert_allclose(wmap[y, x], exp, atol=1e-7)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference checkout not mounted")
def test_reference_pose_optimizer_constructor_runs_unmodified(lib, dataset, monkeypatch):
    """The reference's own pose_optimization.PoseOptimizer.__init__ (reference pose_optimization.py:99-175) against
    this module: only cv2 (two integer constants) is stubbed because OpenCV is not installed here."""
    v, base = dataset
    cv2 = types.ModuleType("cv2")
    cv2.CV_32FC3, cv2.CV_8UC1 = 21, 0
    monkeypatch.setitem(sys.modules, "cv2", cv2)
    monkeypatch.syspath_prepend("/root/reference")
    sys.modules.pop("pose_optimization", None)
    try:
        po = importlib.import_module("pose_optimization")
    except ImportError as e:
        pytest.skip(f"reference helper import failed: {e}")
    opt = types.SimpleNamespace(
        max_iterations=1000, num_threads=12, num_steps=4, robustness=0.5, static_loss_type="ReproDisparity",
        static_spatial_weight=1.0, static_depth_weight=1.0, smooth_loss_type="ReproDisparityLaplacian",
        smooth_static_weight=0.0, smooth_dynamic_weight=0.0, position_regularization=0.0, scale_regularization=1.0,
        scale_regularization_grid_size=10, deformation_regularization_initial=1.0, deformation_regularization_final=0.1,
        adaptive_deformation_cost=0.0, spatial_deformation_regularization=1.0, graduate_deformation_regularization=False,
        focal_regularization=1.0, coarse_to_fine=True, ctf_long=17, ctf_short=10, deferred_spatial_opt=False, dso_long=4,
        dso_short=3, focal_long=0.3461538376301239, intr_opt="PerFrame", fix_poses=False, fix_depth_transforms=False,
        fix_spatial_transforms=False, use_global_scale=False, dynamic_constraints="Mask", epipolar_dist_thresh=1.0)
    p = po.PoseOptimizer(base, "midas2", list(range(v.num_frames)), opt)
    assert p.depth_video.numFrames() == 10 and p.opt_params.ctfLong == 17
    assert os.path.exists(os.path.join(base, "video.dat"))
    sys.modules.pop("pose_optimization", None)


@pytest.mark.gpu
def test_optimize_poses_sequence_on_gpu_matches_direct_c_abi(lib, tmp_path):
    """The reference's optimize_poses() call sequence through lib_python equals driving the C ABI directly."""
    from robust_cvd_amd import api
    from robust_cvd_amd.ctypes_types import OptParams, XformDesc
    from tests.drop_in_caller import build_pose_optimizer, optimize_poses
    v = synth.make_video(10, 96, 56, seed=53)
    base = dataset_io.write_dataset(str(tmp_path / "video"), v)
    frames = list(range(v.num_frames))
    opt = lib.DepthVideoPoseOptimizer.Params()
    opt.ctfLong, opt.ctfShort = 6, 4
    dv, fc = build_pose_optimizer(lib, base, "midas2", frames, opt)
    # initial FOV of the reference pipeline: createDepthStream -> resolveMissingFov defaults (no resetPoses call)
    fov0 = [(dv.depthStream(0).frame(f).intrinsics.vFov, dv.depthStream(0).frame(f).intrinsics.hFov) for f in frames]
    assert abs(fov0[0][0] - 0.666488587) < 1e-6
    optimize_poses(lib, dv, fc, frames, opt)
    ds = dv.depthStream(0)
    assert ds.depthXformDesc().str() == "Grid(Scale, Linear, 6, 4, 1)"
    s = api.Solver(0)
    depth_from_disk = np.stack([np.asarray(ds.frame(f).sourceDepth()) for f in frames])
    s.set_video(v.num_frames, v.width, v.height, dv.aspect(), dv.invAspect())
    s.set_depth_all(depth_from_disk)
    s.set_pair_constraints(v.pairs, v.offsets, v.loc, None)
    s.set_poses(np.zeros((len(frames), 3)), np.tile([0, 0, 0, 1.0], (len(frames), 1)), [a for a, _ in fov0], [b for _, b in fov0])
    p = OptParams.defaults()
    p.ctf_long, p.ctf_short = 6, 4
    p.set_frame_range(frames)
    s.reset_depth_xforms(XformDesc.global_depth())
    s.reset_spatial_xforms(XformDesc.spatial())
    s.normalize_depth(p)
    s.pose_optimization(p)
    poses = s.get_poses()
    th = s.get_xform_params()
    # Nothing pins the global rigid transform (SURVEY.md 7 "gauge freedom"): two solves of the same problem may drift
    # apart along it by far more than their rounding differences, so poses are compared gauge-aligned; the depth
    # transform parameters and the field of view are gauge invariant and compared directly.
    pos = np.stack([np.asarray(ds.frame(f).extrinsics.position) for f in frames])
    quat = np.stack([np.asarray(ds.frame(f).extrinsics.orientation.coeffs()) for f in frames])
    perr, rerr = synth.relative_pose_error(pos, quat, poses["position"], poses["orientation"])
    assert perr < 1e-4 and rerr < 1e-4, (perr, rerr, np.abs(pos - poses["position"]).max())
    for f in frames:
        fr = ds.frame(f)
        assert abs(fr.intrinsics.vFov - poses["vfov"][f]) < 1e-6
        np.testing.assert_allclose(fr.depthXform().params(), th[f], rtol=1e-5)
    # what loaders/video_dataset.py:update_poses reads afterwards
    fr = ds.frame(4)
    assert len(fr.extrinsics.right()) == 3 and np.asarray(fr.depthXform().paramMap(fr)).shape == (56, 96)
    assert np.asarray(fr.spatialXform().warp(ds.height(), ds.width())).shape == (56, 96, 2)
    assert os.path.getsize(os.path.join(base, "video.dat")) > 1000


@pytest.mark.gpu
def test_huber_loss_extension_through_lib_python(lib, tmp_path):
    """Params.huberLoss (extension, off by default) reaches the solver: same result as the C ABI with robust_loss = 1,
    and a different one from the default Cauchy solve."""
    from robust_cvd_amd import api
    from robust_cvd_amd.ctypes_types import OptParams, XformDesc
    from tests.drop_in_caller import build_pose_optimizer, optimize_poses
    v = synth.make_video(8, 96, 56, seed=57, flow_noise_px=1.0)
    frames = list(range(v.num_frames))
    out = {}
    for name, huber in (("cauchy", False), ("huber", True)):
        base = dataset_io.write_dataset(str(tmp_path / name), v)
        opt = lib.DepthVideoPoseOptimizer.Params()
        assert opt.huberLoss is False
        opt.coarseToFine = False
        opt.robustness = 0.01
        opt.huberLoss = huber
        dv, fc = build_pose_optimizer(lib, base, "midas2", frames, opt)
        fov0 = [(dv.depthStream(0).frame(f).intrinsics.vFov, dv.depthStream(0).frame(f).intrinsics.hFov) for f in frames]
        optimize_poses(lib, dv, fc, frames, opt)
        ds = dv.depthStream(0)
        out[name] = np.array([ds.frame(f).depthXform().params() for f in frames])
        if huber:
            s = api.Solver(0)
            s.set_robust_loss(1)
            s.set_video(v.num_frames, v.width, v.height, dv.aspect(), dv.invAspect())
            s.set_depth_all(np.stack([np.asarray(ds.frame(f).sourceDepth()) for f in frames]))
            s.set_pair_constraints(v.pairs, v.offsets, v.loc, None)
            s.set_poses(np.zeros((len(frames), 3)), np.tile([0, 0, 0, 1.0], (len(frames), 1)), [a for a, _ in fov0],
                        [b for _, b in fov0])
            p = OptParams.defaults()
            p.coarse_to_fine = 0
            p.robustness = 0.01
            p.set_frame_range(frames)
            s.reset_depth_xforms(XformDesc.global_depth())
            s.reset_spatial_xforms(XformDesc.spatial())
            s.normalize_depth(p)
            s.pose_optimization(p)
            np.testing.assert_allclose(out[name], s.get_xform_params(), rtol=1e-5)
    assert np.abs(out["huber"] - out["cauchy"]).max() > 1e-4 * np.abs(out["cauchy"]).max()


@pytest.mark.gpu
def test_constraints_are_sampled_from_the_flow_images_on_the_gpu(lib, tmp_path):
    """FlowConstraintsCollection(video, params) without a cache file: compute() + save() (reference
    lib/FlowConstraints.cpp:84-93, 288-550) through the device kernels; the cache it writes must hold exactly what the
    oracle's chain (corner response -> dynamic distance -> greedy sampling) gives on the same images."""
    from oracle.oracle import Oracle
    F, W, H, sep = 4, 64, 40, 6
    v = synth.make_video(F, W, H, seed=53, spacing=8)
    base = dataset_io.write_dataset(str(tmp_path / "v"), v)
    os.remove(os.path.join(base, "flow_constraints.dat"))
    rng = np.random.default_rng(11)
    pairs = np.array([[0, 1], [1, 0], [1, 2], [2, 1], [2, 3], [3, 2]], np.int32)
    flows = rng.normal(0, 1.5, (len(pairs), H, W, 2)).astype(np.float32)
    masks = np.where(rng.uniform(size=(len(pairs), H, W)) < 0.9, 255, 0).astype(np.uint8)
    colors = rng.uniform(0, 1, (F, H, W, 3)).astype(np.float32)
    dyn = np.full((F, H // 2, W // 2), 255, np.uint8)
    dyn[:, 4:8, 6:14] = 0
    with open(os.path.join(base, "flow_list.json"), "w") as f:
        import json
        json.dump([["src", "dst"]] + pairs.tolist(), f)
    dataset_io.write_flow_inputs(base, pairs, flows, masks, colors, dyn)
    dv = lib.DepthVideo()
    lib.DepthVideoImporter.importVideo(dv, base, True)
    assert dv.hasColorStream("down") and dv.hasColorStream("dynamic_mask")
    fcp = lib.FlowConstraintsParams()
    fcp.frameRange.resolve(dv.numFrames(), True)
    fcp.matchSeparation = sep
    fcp.minDynamicDistance = 2
    fc = lib.FlowConstraintsCollection(dv, fcp)          # no cache: compute + save
    assert os.path.exists(os.path.join(base, "flow_constraints.dat")) and fc.numConstraints() > 40
    got_sep, got_pairs, got_trips = dataset_io.read_flow_constraints(os.path.join(base, "flow_constraints.dat"), len(pairs), F - 2)
    assert got_sep == sep
    o = Oracle()
    synth.load_into(o, v)
    corner = o.corner_min_eigenval(colors)
    dd = o.dynamic_distance(dyn)
    off, loc = o.sample_pair_constraints(pairs, corner, flows, masks, sep, dyn_dist=dd, min_dynamic_distance=2.0)
    for k, (a, b) in enumerate(pairs.tolist()):
        assert np.array_equal(got_pairs[(a, b)], loc[off[k]:off[k + 1]])
    idx = {tuple(p): k for k, p in enumerate(pairs.tolist())}
    centers = np.array([1, 2], np.int32)
    f10 = np.stack([flows[idx[(c, c - 1)]] for c in centers]); m10 = np.stack([masks[idx[(c, c - 1)]] for c in centers])
    f12 = np.stack([flows[idx[(c, c + 1)]] for c in centers]); m12 = np.stack([masks[idx[(c, c + 1)]] for c in centers])
    toff, tloc = o.sample_triplet_constraints(centers, corner, f10, m10, f12, m12, sep, dyn_dist=dd, min_dynamic_distance=2.0)
    for k, c in enumerate(centers.tolist()):
        assert np.array_equal(got_trips[c], tloc[toff[k]:toff[k + 1]])
    # a second construction finds the cache (same counts), and the dynamic-mask flags follow the distance map
    fc2 = lib.FlowConstraintsCollection(dv, fcp)
    assert fc2.numConstraints() == fc.numConstraints()
    fc2.setStaticFlagFromDynamicMask(3)
    fc2.resetStaticFlag()


@pytest.mark.gpu
def test_match_separation_zero_goes_to_the_solver_as_images(lib, tmp_path, monkeypatch):
    """FlowConstraintsParams.matchSeparation = 0 (reference lib/FlowConstraints.cpp:315-329, 381-395: every masked in-bounds
    pixel is a constraint): the collection keeps the flow / mask images it was computed from and DepthVideoProcessor hands
    THOSE to the solver (cvd_set_pair_flows, dense mode) instead of the materialised list.  Same optimize_poses() sequence
    with the hand-over switched off (lib_python.setDenseHandOver(False): the list path) must end in the same state."""
    import json
    from tests.drop_in_caller import build_pose_optimizer, optimize_poses
    v = synth.make_video(8, 96, 56, seed=71)
    flows, masks = synth.make_dense_flows(v)
    F, H, W = v.num_frames, v.height, v.width
    colors = np.random.default_rng(5).uniform(0, 1, (F, H, W, 3)).astype(np.float32)
    frames = list(range(F))
    res = {}
    for mode in ("images", "list"):
        base = dataset_io.write_dataset(str(tmp_path / mode), v)
        os.remove(os.path.join(base, "flow_constraints.dat"))
        with open(os.path.join(base, "flow_list.json"), "w") as f:
            json.dump([["src", "dst"]] + v.pairs.tolist(), f)
        dataset_io.write_flow_inputs(base, v.pairs, flows, masks, colors)
        lib.setDenseHandOver(mode != "list")
        opt = lib.DepthVideoPoseOptimizer.Params()
        opt.ctfLong, opt.ctfShort = 6, 4
        # (tests/drop_in_caller.build_pose_optimizer with matchSeparation = 0 and no dynamic-mask flags)
        from tests.drop_in_caller import CV_32FC3
        dv = lib.DepthVideo()
        lib.DepthVideoImporter.importVideo(dv, base, False)
        dv.createColorStream("full", "color_full", ".png", CV_32FC3)
        dv.createColorStream("down", "color_down", ".raw", CV_32FC3)
        dv.createDepthStream("depth_midas2", "depth_midas2", [-1, -1])
        dv.save()
        fcp = lib.FlowConstraintsParams()
        fcp.frameRange.resolve(dv.numFrames(), True)
        fcp.matchSeparation = 0
        fc = lib.FlowConstraintsCollection(dv, fcp)        # no cache: compute (+ save)
        assert fc.holdsFlowImages()
        n_valid = int(synth.dense_constraints_from_flows(v, flows, masks)[0][-1])
        assert fc.numConstraints() == n_valid             # the materialised list: every valid pixel, as the reference would
        proc = optimize_poses(lib, dv, fc, frames, opt)
        assert proc.usedFlowImages == (mode == "images")
        ds = dv.depthStream(dv.numDepthStreams() - 1)
        res[mode] = (np.stack([np.asarray(ds.frame(f).extrinsics.position) for f in frames]),
                     np.stack([np.asarray(ds.frame(f).extrinsics.orientation.coeffs()) for f in frames]),
                     np.stack([np.asarray(ds.frame(f).depthXform().params()) for f in frames]),
                     ds.depthXformDesc().str())
        lib.setDenseHandOver(True)   # (module-wide switch: back to its default for the tests that follow)
    assert res["images"][3] == res["list"][3] == "Grid(Scale, Linear, 6, 4, 1)"
    perr, rerr = synth.relative_pose_error(res["images"][0], res["images"][1], res["list"][0], res["list"][1])
    # (two paths -- image-reading kernels with explicit blocks, list kernels matrix-free -- through the whole default schedule at
    # eta = 1e-3: typically 1e-5 - 3e-5 apart, ~1e-4 when a function-tolerance test falls the other way at one level;
    # the bar for "the same end state" is the parity bar's third)
    from tests import margins
    margins.below("position images vs list", perr, 3e-4)
    margins.below("rotation images vs list", rerr, 3e-4)
    margins.below("depth parameters images vs list", float(np.max(np.abs(res["images"][2] - res["list"][2]) / np.abs(res["list"][2]))), 3e-4)


@pytest.mark.gpu
def test_match_separation_zero_outside_the_dense_scope_takes_the_list(lib, tmp_path):
    """A matchSeparation = 0 collection with a residual configuration the dense mode does not cover (here: the Euclidean loss; the
    same holds for cubic grids, spatial transforms, smoothness, shared intrinsics, ScaleShift) must go to the solver as the
    materialised LIST, as it did before the image hand-over existed -- not fail in the library's scope check
    (cvd_dense_mode_supported)."""
    import json
    from tests.drop_in_caller import CV_32FC3, optimize_poses
    v = synth.make_video(6, 64, 40, seed=72)
    flows, masks = synth.make_dense_flows(v)
    F, H, W = v.num_frames, v.height, v.width
    colors = np.random.default_rng(6).uniform(0, 1, (F, H, W, 3)).astype(np.float32)
    base = dataset_io.write_dataset(str(tmp_path / "v"), v)
    os.remove(os.path.join(base, "flow_constraints.dat"))
    with open(os.path.join(base, "flow_list.json"), "w") as f:
        json.dump([["src", "dst"]] + v.pairs.tolist(), f)
    dataset_io.write_flow_inputs(base, v.pairs, flows, masks, colors)
    dv = lib.DepthVideo()
    lib.DepthVideoImporter.importVideo(dv, base, False)
    dv.createColorStream("full", "color_full", ".png", CV_32FC3)
    dv.createColorStream("down", "color_down", ".raw", CV_32FC3)
    dv.createDepthStream("depth_midas2", "depth_midas2", [-1, -1])
    dv.save()
    fcp = lib.FlowConstraintsParams()
    fcp.frameRange.resolve(dv.numFrames(), True)
    fcp.matchSeparation = 0
    fc = lib.FlowConstraintsCollection(dv, fcp)
    assert fc.holdsFlowImages()
    opt = lib.DepthVideoPoseOptimizer.Params()
    opt.ctfLong, opt.ctfShort = 4, 3
    opt.staticLossType = lib.StaticLossType.Euclidean
    proc = optimize_poses(lib, dv, fc, list(range(F)), opt)   # (raised "dense mode ... supports" before)
    assert not proc.usedFlowImages
    ds = dv.depthStream(dv.numDepthStreams() - 1)
    assert ds.depthXformDesc().str() == "Grid(Scale, Linear, 4, 3, 1)"
    assert np.isfinite(np.stack([np.asarray(ds.frame(f).extrinsics.position) for f in range(F)])).all()
    # An Identity depth transform (the stream's state before any reset: no value parameter, N = 0) is outside the dense scope
    # as well (cvd_dense_mode_supported answered 1 and the solve died in checkDenseScope): poses only, as a list.
    params = lib.DepthVideoProcessor.Params()
    params.depthStream = dv.numDepthStreams() - 1
    params.frameRange.fromString(",".join(str(x) for x in range(F)))
    params.poseOptimizer = lib.DepthVideoPoseOptimizer.Params()
    params.poseOptimizer.frameRange.fromString(",".join(str(x) for x in range(F)))
    params.poseOptimizer.coarseToFine = False
    params.poseOptimizer.numSteps = 1
    params.op = lib.DepthVideoProcessor.Op.ResetDepthXforms
    params.depthXformDesc.type = lib.XformType.Depth
    params.depthXformDesc.depthType = lib.DepthXformType.Identity
    proc.process(params)
    assert ds.depthXformDesc().str().startswith("Identity")
    proc.optimizePoses(params, fc)
    assert not proc.usedFlowImages
    assert np.isfinite(np.stack([np.asarray(ds.frame(f).extrinsics.position) for f in range(F)])).all()


@pytest.mark.gpu
def test_flow_guided_filter_op_matches_the_oracle(lib, tmp_path):
    """filter_depth() of the reference (pose_optimization.py:295-325): Op.Copy then Op.FlowGuidedFilter with
    frameRadius = radius, through the files of the dataset; result = the oracle's filter on the same arrays."""
    from oracle.oracle import Oracle
    from tests.filter_cases import make_case
    F, W, H, R = 6, 48, 28, 2
    v = synth.make_video(F, W, H, seed=54, spacing=8)
    base = dataset_io.write_dataset(str(tmp_path / "v"), v)
    c = make_case(F, W, H, seed=21)
    pairs = [[k, k + 1] for k in range(F - 1)] + [[k + 1, k] for k in range(F - 1)]
    flows = list(c["flow_fwd"]) + list(c["flow_bwd"])
    masks = list(c["mask_fwd"]) + list(c["mask_bwd"])
    dataset_io.write_flow_inputs(base, pairs, flows, masks, np.zeros((F, H, W, 3), np.float32))
    dv = lib.DepthVideo()
    lib.DepthVideoImporter.importVideo(dv, base, True)
    dv.createDepthStream("depth_midas2", "depth_midas2", [W, H])
    src = dv.depthStream(0)
    for f in range(F):
        fr = src.frame(f)
        fr.setDepth(c["depth"][f])
        e = fr.extrinsics
        e.position = [float(x) for x in c["cameras"][f, :3]]
        q = e.orientation
        q.setCoeffs([float(x) for x in c["cameras"][f, 3:7]])
        e.orientation = q
        fr.extrinsics = e
        i = fr.intrinsics
        i.hFov, i.vFov = float(c["cameras"][f, 7]), float(c["cameras"][f, 8])
        fr.intrinsics = i
    dv.createDepthStream("filtered", "depth_filtered", [W, H])
    proc = lib.DepthVideoProcessor(dv)
    params = lib.DepthVideoProcessor.Params()
    params.frameRange.fromString("1-4")
    params.op = lib.DepthVideoProcessor.Op.Copy
    params.sourceDepthStream, params.depthStream = 0, 1
    proc.process(params)
    assert np.array_equal(dv.depthStream(1).frame(2).depth(), c["depth"][2])
    params.op = lib.DepthVideoProcessor.Op.FlowGuidedFilter
    params.frameRadius = R
    proc.process(params)
    got = np.stack([dv.depthStream(1).frame(f).depth() for f in range(1, 5)])
    inv_aspect = np.float32(dv.invAspect())
    o = Oracle()
    # batch = frames max(0, 1 - R) .. 4 = 0 .. 4, outputs 1 .. 4
    ref = o.flow_guided_filter(c["depth"][:5], c["cameras"][:5], c["flow_fwd"][:4], c["mask_fwd"][:4], c["flow_bwd"][:4],
                               c["mask_bwd"][:4], inv_aspect, R, first=1, count=4)
    assert np.allclose(got, ref, rtol=2e-6, atol=0)
    params.farConnections = True
    with pytest.raises(RuntimeError, match="farConnections"):
        proc.process(params)
