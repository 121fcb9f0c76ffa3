#!/usr/bin/env python3
"""Mint tests/golden/reference_py/reprojection_golden.npz: the reference's OWN consumer code run on the drop-in's outputs.

Two phases, because the optimizer needs the GPU and the reference checkout (/root/reference) only exists in the build
container:

    # 1. on the GPU box (through gpurun): the drop-in's optimize_poses on the seeded zero-noise case, every getter dumped
    python tests/golden/reference_py/make_reprojection_golden.py dump gpurun_out/reproj_dump.npz
    # 2. in the container: the dumped values replayed through the REFERENCE's VideoDataset.update_poses
    #    (loaders/video_dataset.py:153-217) and utils/geometry.py:62-138; what they return is committed
    python tests/golden/reference_py/make_reprojection_golden.py mint gpurun_out/reproj_dump.npz

The replay object answers exactly the calls update_poses makes (numFrames, numDepthStreams, depthStream(i).frame(j)
.extrinsics.right() ... .depthXform().paramMap(frame), .spatialXform().warp(h, w)) with the values the drop-in's getters
returned on the GPU box; the enum classes the reference compares against are the drop-in module's own (the reference
imports them `from lib_python`).  Only `cv2` is stubbed (OpenCV is not installed here; update_poses does not use it).
"""
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)

from tests import reference_reprojection as rr  # noqa: E402


def _lib():
    import importlib
    from robust_cvd_amd import build as b
    d = os.path.dirname(b.build_lib_python())
    if d not in sys.path:
        sys.path.insert(0, d)
    return importlib.import_module("lib_python")


def dump(path):
    lib = _lib()
    video = rr.make_case()
    with tempfile.TemporaryDirectory() as tmp:
        out = rr.run_drop_in(lib, video, os.path.join(tmp, "video"))
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savez_compressed(path, **out)
    print("dumped", path, bytes(out["depth_desc"]).decode())


class _Replay:
    """The calls VideoDataset.update_poses makes on a lib_python.DepthVideo, answered from the dump."""

    def __init__(self, lib, out):
        self.lib, self.out = lib, out

    def numFrames(self):
        return int(self.out["position"].shape[0])

    def numDepthStreams(self):
        return 1

    def depthStream(self, i):
        assert i == 0
        return self

    def width(self):
        return int(self.out["width"])

    def height(self):
        return int(self.out["height"])

    def frame(self, i):
        lib, out = self.lib, self.out
        desc = bytes(out["depth_desc"]).decode()
        ext = types.SimpleNamespace(right=lambda: out["right"][i].tolist(), up=lambda: out["up"][i].tolist(),
                                    backward=lambda: out["backward"][i].tolist(), position=out["position"][i].tolist())
        intr = types.SimpleNamespace(hFov=float(out["hfov"][i]), vFov=float(out["vfov"][i]))
        ddesc = types.SimpleNamespace(
            type=lib.XformType.Depth, valueXform=lib.ValueXformType.Scale,
            depthType={"Identity": lib.DepthXformType.Identity, "Global": lib.DepthXformType.Global,
                       "Grid": lib.DepthXformType.Grid}[desc.split("(")[0]])
        dx = types.SimpleNamespace(desc=lambda: ddesc, params=lambda: out["params"][i].tolist(),
                                   paramMap=lambda f: out["param_map"][i])
        sdesc = types.SimpleNamespace(type=lib.XformType.Spatial, spatialType=lib.SpatialXformType.Identity)
        sx = types.SimpleNamespace(desc=lambda: sdesc, warp=lambda h, w: out["warp"][i])
        return types.SimpleNamespace(extrinsics=ext, intrinsics=intr, depthXform=lambda: dx, spatialXform=lambda: sx)


def reference_outputs(out, video):
    """The reference's update_poses + geometry functions on the dumped values (needs /root/reference)."""
    import torch
    lib = _lib()
    cv2 = types.ModuleType("cv2")
    cv2.CV_32FC3, cv2.CV_8UC1, cv2.IMREAD_UNCHANGED = 21, 0, -1
    sys.modules.setdefault("cv2", cv2)
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    from loaders.video_dataset import VideoDataset  # the reference's class
    from utils import geometry                      # the reference's camera model
    self_like = types.SimpleNamespace(frames=list(range(video.num_frames)))
    VideoDataset.update_poses(self_like, _Replay(lib, out))
    ext = self_like.extrinsics.numpy()
    intr = self_like.intrinsics.numpy()
    scales = np.stack([self_like.scales[i].numpy() for i in self_like.frames])
    warp = np.stack([self_like.warp_map[i].numpy() for i in self_like.frames])
    fa, fb, pix, target, depth = rr.constraint_samples(video, {**out, "param_map": scales.astype(np.float64)})
    # one constraint per batch entry: (B, C, 1, 1) tensors through the reference's functions
    B = len(fa)
    t = lambda a: torch.tensor(a, dtype=torch.float32)
    pixels = t(pix).view(B, 2, 1, 1)
    points = geometry.pixels_to_points(t(intr[fa]), t(depth).view(B, 1, 1, 1), pixels.clone())
    world = geometry.points_cam_to_world(points, t(ext[fa]))
    cam_b = geometry.world_to_points_cam(world, t(ext[fb]))
    via_reproject = geometry.reproject_points(points, t(ext[fa]), t(ext[fb]))
    assert torch.allclose(cam_b, via_reproject, atol=1e-5)
    reproj = geometry.project(cam_b, t(intr[fb])).view(B, 2).numpy()
    return dict(ref_extrinsics=ext, ref_intrinsics=intr, ref_scales=scales, ref_warp_abs_max=np.float32(np.abs(warp).max()),
                ref_reprojected=reproj, frame_a=fa.astype(np.int32), frame_b=fb.astype(np.int32), source_pixel=pix, target_pixel=target,
                source_depth_scaled=depth)


def mint(path):
    out = dict(np.load(path))
    video = rr.make_case()
    ref = reference_outputs(out, video)
    err = np.linalg.norm(ref["ref_reprojected"] - ref["target_pixel"], axis=1)
    spread, worst = rr.depth_scale_spread(video, {**out, "param_map": ref["ref_scales"].astype(np.float64)})
    print(f"{len(err)} static constraints: reprojection error through the reference's geometry.py max {err.max():.4f} px, "
          f"mean {err.mean():.4f} px; per-frame scale spread {spread:.2e}, worst pixel {worst:.2e}")
    # what the pin can see: the same state with paramMap's rows in the wrong order (row 0 = image bottom)
    flipped = reference_outputs({**out, "param_map": out["param_map"][:, ::-1].copy()}, video)
    err_flip = np.linalg.norm(flipped["ref_reprojected"] - flipped["target_pixel"], axis=1)
    print(f"with paramMap flipped vertically the same check gives max {err_flip.max():.4f} px, mean {err_flip.mean():.4f} px")
    ref["flipped_rows_error_px_max"] = np.float32(err_flip.max())
    keep = {k: out[k] for k in ("right", "up", "backward", "position", "orientation", "hfov", "vfov", "params", "width", "height",
                                "depth_desc")}
    # (the full-resolution maps of three frames only: the fixture stays small; every constraint's scaled depth is kept)
    sub = np.array([0, video.num_frames // 2, video.num_frames - 1])
    keep["map_frames"] = sub.astype(np.int32)
    keep["param_map"] = out["param_map"][sub]
    ref["ref_scales"] = ref["ref_scales"][sub]
    from tests import baseline_configs as bc
    np.savez_compressed(rr.GOLDEN, input_sha256=np.frombuffer(bc.input_digest(video).encode(), np.uint8),
                        reprojection_error_px=err.astype(np.float32), **keep, **ref)
    print("wrote", rr.GOLDEN, os.path.getsize(rr.GOLDEN), "bytes")


if __name__ == "__main__":
    {"dump": dump, "mint": mint}[sys.argv[1]](sys.argv[2])
