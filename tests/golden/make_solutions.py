#!/usr/bin/env python3
"""Mint tests/golden/solutions/*.npz: the CPU oracle's END STATE for BASELINE.json configs[0..2] at their real sizes
(run from the repo root; config2 takes ~1.5 min on 8 cores).

    python tests/golden/make_solutions.py [config0 config1 config2]

The oracle (oracle/cvd_oracle.cpp: dual-number autodiff, Ceres-default LM, EXACT block-sparse Cholesky on the frame
graph) is a CPU restatement of the reference's Ceres solve -- parity unpinned against a real Ceres build, see
DESIGN.md.  The synthetic inputs are regenerated from the seed on the GPU box (tests/baseline_configs.py); the file
keeps their SHA-256 so that a drifting generator is noticed.  Besides the default-tolerance solution (what Ceres'
function_tolerance = 1e-6 stops at) a tightly converged one (function_tolerance 1e-12, started from the former) is
stored: the distance between the two is the resolution at which "the reference's result" is defined at all.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.oracle import Oracle  # noqa: E402
from robust_cvd_amd import synth  # noqa: E402
from tests import baseline_configs as bc  # noqa: E402


def main():
    names = sys.argv[1:] or list(bc.CONFIGS)
    os.makedirs(bc.SOLUTIONS_DIR, exist_ok=True)
    threads = min(12, os.cpu_count() or 1)
    for name in names:
        video = bc.make_video(name)
        o = Oracle()
        t0 = time.time()
        sol = bc.run(o, name, video, threads)
        dt = time.time() - t0
        # tightly converged minimum of the LAST level, continued from the default-tolerance end state
        o.set_function_tolerance(1e-12)
        p = bc.params_for(name, threads)
        p.max_iterations = 50
        o.pose_optimization_step(p, p.depth_deform_reg_final, convert_poses=False)
        poses = o.get_poses()
        tight = {"pose7": o.get_pose_params().copy(), "position": poses["position"].copy(),
                 "orientation": poses["orientation"].copy(), "depth_params": o.get_xform_params().copy(),
                 "final_cost": o.summary()["final_cost"], "iterations": o.summary()["num_iterations"]}
        perr, rerr = synth.relative_pose_error(sol["position"], sol["orientation"], tight["position"], tight["orientation"])
        s = sol["summary"]
        ncons = int(video.dense_offsets[-1]) if bc.CONFIGS[name].get("dense") else video.num_constraints
        print(f"{name}: {video.num_frames} frames, {len(video.pairs)} pairs, {ncons} constraints; oracle "
              f"{dt:.1f} s, final cost {s['final_cost']:.9f} ({s['num_iterations']} LM iterations in the last level); tight "
              f"{tight['final_cost']:.9f} after {tight['iterations']} more; default-vs-tight pose err {perr:.2e} rot {rerr:.2e} "
              f"params {np.abs(sol['depth_params'] - tight['depth_params']).max() / np.abs(tight['depth_params']).max():.2e}")
        np.savez_compressed(
            bc.solution_path(name), input_sha256=np.frombuffer(bc.input_digest(video).encode(), np.uint8),
            num_pairs=len(video.pairs),
            num_constraints=(int(video.dense_offsets[-1]) if bc.CONFIGS[name].get("dense") else video.num_constraints),
            pose7=sol["pose7"], position=sol["position"], orientation=sol["orientation"], vfov=sol["vfov"], hfov=sol["hfov"],
            depth_params=sol["depth_params"], grid_size=sol["grid_size"], final_cost=s["final_cost"],
            initial_cost_last_level=s["initial_cost"], iterations_last_level=s["num_iterations"], oracle_seconds=dt,
            tight_pose7=tight["pose7"], tight_position=tight["position"], tight_orientation=tight["orientation"],
            tight_depth_params=tight["depth_params"], tight_final_cost=tight["final_cost"])


if __name__ == "__main__":
    main()
