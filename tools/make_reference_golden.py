#!/usr/bin/env python3
"""Mint REAL-reference golden vectors: run facebookresearch/robust_cvd's own pose optimizer (its `lib_python` built against
Ceres, on any machine that has one) on a synthetic dataset written by robust_cvd_amd.dataset_io and keep its `video.dat`.

    python tools/make_reference_golden.py --reference /path/to/robust_cvd --lib-python /path/to/robust_cvd/lib/build \\
        [--config config0] [--out tests/golden/reference_ceres]

What it does
  1. regenerates the seeded synthetic video of tests/baseline_configs.py (`--config`, default config0: 30 frames 192x112)
     and writes it as a dataset directory: frames.txt, depth_<model>/depth/frame_%06d.raw (disparity), flow_list.json,
     flow_constraints.dat (the constraint cache the reference loads instead of sampling, lib/FlowConstraints.cpp:116-189),
     empty colour stream directories;
  2. imports the REFERENCE's `pose_optimization.py` with the REFERENCE's compiled `lib_python` on sys.path and runs
     `PoseOptimizer(base, model, frames, opt).optimize_poses()` with the stock `--opt.*` defaults (params.py:96-190);
  3. copies the resulting `video.dat` (poses, FOV, depth-transform parameters: lib/DepthVideo.cpp:300-385) and a small
     metadata file to `--out/<config>/`.
tests/test_reference_golden.py then compares this repository's solve of the same inputs against it (GPU) and pins the
oracle against it (CPU) -- the route from "parity unpinned" to a reference-anchored oracle.  Nothing here runs on the GPU
box or in CI unless such a file has been committed; this container has neither Ceres nor the reference's build.
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True, help="checkout of facebookresearch/robust_cvd")
    ap.add_argument("--lib-python", required=True, help="directory holding the reference's compiled lib_python*.so")
    ap.add_argument("--config", default="config0")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "reference_ceres"))
    ap.add_argument("--model-type", default="midas2")
    args = ap.parse_args()

    from robust_cvd_amd import dataset_io
    from tests import baseline_configs as bc

    video = bc.make_video(args.config)
    work = tempfile.mkdtemp(prefix="cvd_ref_")
    base = dataset_io.write_dataset(os.path.join(work, "video"), video, model_type=args.model_type)

    sys.path.insert(0, args.lib_python)  # the reference's lib_python (NOT robust_cvd_amd/lib)
    sys.path.insert(0, args.reference)
    import lib_python  # noqa: F401
    if "robust_cvd_amd" in os.path.abspath(lib_python.__file__):
        raise SystemExit("--lib-python points at this repository's module, not at the reference's build")
    import pose_optimization

    dflt = lib_python.DepthVideoPoseOptimizer.Params()
    p = bc.params_for(args.config)
    opt = types.SimpleNamespace(
        max_iterations=dflt.maxIterations, num_threads=dflt.numThreads, num_steps=p.num_steps, robustness=dflt.robustness,
        static_loss_type="ReproDisparity", static_spatial_weight=1.0, static_depth_weight=1.0,
        smooth_loss_type="ReproDisparityLaplacian", smooth_static_weight=0.0, smooth_dynamic_weight=0.0,
        position_regularization=0.0, scale_regularization=1.0, scale_regularization_grid_size=10,
        deformation_regularization_initial=1.0, deformation_regularization_final=0.1, adaptive_deformation_cost=0.0,
        spatial_deformation_regularization=1.0, graduate_deformation_regularization=False, focal_regularization=1.0,
        coarse_to_fine=bool(p.coarse_to_fine), ctf_long=p.ctf_long, ctf_short=p.ctf_short, deferred_spatial_opt=False,
        dso_long=4, dso_short=3, focal_long=dflt.focalLong,
        intr_opt={0: "Fixed", 1: "Shared", 2: "PerFrame"}[int(p.intr_opt)], fix_poses=False, fix_depth_transforms=False,
        fix_spatial_transforms=False, use_global_scale=False, dynamic_constraints="None", epipolar_dist_thresh=1.0)
    po = pose_optimization.PoseOptimizer(base, args.model_type, list(range(video.num_frames)), opt)
    po.optimize_poses()

    out = os.path.join(args.out, args.config)
    os.makedirs(out, exist_ok=True)
    shutil.copy(os.path.join(base, "video.dat"), os.path.join(out, "video.dat"))
    with open(os.path.join(out, "meta.json"), "w") as f:
        json.dump({"config": args.config, "input_sha256": bc.input_digest(video), "frames": video.num_frames,
                   "pairs": int(len(video.pairs)), "constraints": int(video.num_constraints), "model_type": args.model_type,
                   "note": "config1's bicubic 4x4 grid is not reachable from the reference's Python (cubicInterpolation is not "
                           "bound): config0 and config2 are the configurations to mint"}, f, indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
