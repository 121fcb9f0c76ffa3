#!/usr/bin/env python3
"""GPU: run-to-run variation of a full coarse-to-fine solve (same inputs, fresh handle each time)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robust_cvd_amd import api, synth
from robust_cvd_amd.ctypes_types import OptParams, XformDesc

frames, w, h, seed, runs = (int(a) for a in (sys.argv[1:6] + ["10", "96", "56", "53", "6"][len(sys.argv) - 1:]))
eta = float(os.environ.get("ETA", "0")) or None
v = synth.make_video(frames, w, h, seed=seed)
first = None
for r in range(runs):
    s = api.Solver(0)
    if eta:
        s.set_options(pcg_relative_tolerance=eta)
    synth.load_into(s, v)
    p = OptParams.defaults()
    if frames <= 30:
        p.ctf_long, p.ctf_short = 6, 4
    s.reset_depth_xforms(XformDesc.global_depth()); s.reset_spatial_xforms(XformDesc.spatial())
    s.normalize_depth(p)
    s.pose_optimization(p)
    recs = s.records()
    sm = s.summary()
    pos = s.get_poses()["position"].astype(np.float64)
    th = s.get_xform_params()
    if first is None:
        first = (pos, th)
    levels, cur = [], []
    for rec in recs:
        if rec["iteration"] == 0 and cur:
            levels.append(cur); cur = []
        cur.append(rec)
    levels.append(cur)
    desc = " | ".join(f"{len(l) - 1} it ({sum(x['linear_iterations'] for x in l)} pcg) last dcost/cost {abs(l[-1]['cost_change']) / l[-1]['cost']:.2e}" for l in levels)
    print(f"run {r}: cost {sm['final_cost']:.15f} LM {sm['num_iterations']} PCG {sm['total_linear_iterations']}  dpos {np.abs(pos - first[0]).max():.2e} dtheta {np.abs(th - first[1]).max():.2e}\n      {desc}", flush=True)
    s.close()
