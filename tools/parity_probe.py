#!/usr/bin/env python3
"""Development aid (GPU): end-state distance to the oracle's minted solutions (tests/golden/solutions) and PCG effort for a
set of solver options.  usage: parity_probe.py "eta=5e-3" "eta=1e-3,coarse_level=0" ... [--configs=config2_4k,config2]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robust_cvd_amd import api, synth
from tests import baseline_configs as bc

args = [a for a in sys.argv[1:] if not a.startswith("--")]
configs = ["config2_4k", "config2", "config1", "config0"]
for a in sys.argv[1:]:
    if a.startswith("--configs="):
        configs = a.split("=", 1)[1].split(",")
for name in configs:
    v = bc.make_video(name)
    ref = bc.load_solution(name)
    for spec in args:
        kv = dict(x.split("=") for x in spec.split(","))
        s = api.Solver(0)
        s.set_options(pcg_relative_tolerance=float(kv.pop("eta", 1e-3)))
        if kv:  # any other cvd_solver_options field, e.g. coarse_level=0
            s.set_options(**{k: (float(v) if "." in v or "e" in v else int(v)) for k, v in kv.items()})
        sol = bc.run(s, name, v)
        sm = sol["summary"]
        perr, rerr = synth.relative_pose_error(sol["position"], sol["orientation"], ref["position"], ref["orientation"])
        th = float(np.abs(sol["depth_params"] - ref["depth_params"]).max() / np.abs(ref["depth_params"]).max())
        fc = float(ref["final_cost"])
        print(f"{name:11s} {spec:24s} perr {perr:.2e} rerr {rerr:.2e} theta {th:.2e} cost rel {abs(sm['final_cost'] - fc) / fc:.2e} "
              f"LM {sm['num_iterations']:3d} PCG {sm['total_linear_iterations']:5d} last-level LM {len(s.records()) - 1} "
              f"(oracle {int(ref['iterations_last_level'])}) {sm['total_seconds'] * 1e3:7.1f} ms", flush=True)
        s.close()
