#!/usr/bin/env python3
"""GPU: end-state parity of the HIP path against the committed oracle solutions (tests/golden/solutions) for the
BASELINE configs at their real sizes, swept over the PCG forcing value eta.

    python tools/parity_sweep.py [config0 config1 config2] [--eta 0.1,0.01,...]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from robust_cvd_amd import api, synth  # noqa: E402
from tests import baseline_configs as bc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=list(bc.CONFIGS))
    ap.add_argument("--eta", default="0.1,0.03,0.01,0.003,0.001")
    ap.add_argument("--extra", default="", help="extra solver options k=v,k=v")
    args = ap.parse_args()
    extra = {}
    for kv in filter(None, args.extra.split(",")):
        k, v = kv.split("=")
        extra[k] = float(v) if "." in v or "e" in v else int(v)
    for name in args.configs:
        video = bc.make_video(name)
        ref = bc.load_solution(name)
        same = bc.input_digest(video).encode() == ref["input_sha256"].tobytes()
        print(f"== {name}: {video.num_frames} frames {len(video.pairs)} pairs {video.num_constraints} constraints, inputs "
              f"{'identical to' if same else 'DIFFER from'} the minted ones; oracle final cost {float(ref['final_cost']):.9f} "
              f"({int(ref['iterations_last_level'])} LM it. last level), tight {float(ref['tight_final_cost']):.9f}")
        for eta in [float(e) for e in args.eta.split(",")]:
            s = api.Solver(0)
            s.set_options(pcg_relative_tolerance=eta, **extra)
            t0 = time.perf_counter()
            sol = bc.run(s, name, video)
            dt = time.perf_counter() - t0
            sm = sol["summary"]
            recs = s.records()
            perr, rerr = synth.relative_pose_error(sol["position"], sol["orientation"], ref["position"], ref["orientation"])
            perr_t, rerr_t = synth.relative_pose_error(sol["position"], sol["orientation"], ref["tight_position"], ref["tight_orientation"])
            dp = np.abs(sol["depth_params"] - ref["depth_params"]).max() / np.abs(ref["depth_params"]).max()
            dpt = np.abs(sol["depth_params"] - ref["tight_depth_params"]).max() / np.abs(ref["tight_depth_params"]).max()
            fov = np.abs(sol["vfov"] - ref["vfov"]).max()
            print(f"  eta {eta:7.4f}: {dt:6.3f} s  LM {sm['num_iterations']:3d} PCG {sm['total_linear_iterations']:5d}  cost "
                  f"{sm['final_cost']:.9f} (rel {abs(sm['final_cost'] - float(ref['final_cost'])) / float(ref['final_cost']):.1e})  "
                  f"vs oracle: pos {perr:.2e} rot {rerr:.2e} theta {dp:.2e} fov {fov:.1e} | vs tight: pos {perr_t:.2e} rot {rerr_t:.2e} "
                  f"theta {dpt:.2e} | records {len(recs)}", flush=True)
            s.close()


if __name__ == "__main__":
    main()
