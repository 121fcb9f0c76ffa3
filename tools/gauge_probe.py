#!/usr/bin/env python3
"""GPU: the near-gauge case of tests/test_reference_reprojection.py (scaleReg = 1e-6) solved repeatedly through the C ABI -- per
run and coarse-to-fine level the LM iterations, PCG iterations, termination, final cost, and the end state's overall scale (median
depth scale): which level lets two runs of one build part by 20 % in scale while reprojecting equally well?
usage: gauge_probe.py [runs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
from robust_cvd_amd import api, synth
from robust_cvd_amd.ctypes_types import OptParams, XformDesc
from tests import reference_reprojection as rr

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
v = rr.make_case()
for r in range(runs):
    s = api.Solver(0)
    synth.load_into(s, v)
    p = OptParams.defaults()
    p.ctf_long, p.ctf_short = rr.CASE["ctf"]
    p.depth_deform_reg_initial = p.depth_deform_reg_final = rr.CASE["deform_reg"]
    p.scale_reg = rr.CASE["scale_reg"]
    s.reset_depth_xforms(XformDesc.global_depth())
    s.reset_spatial_xforms(XformDesc.spatial())
    s.normalize_depth(p)
    s.pose_optimization(p)
    recs = s.records()
    levels, cur = [], []
    for rec in recs:
        if rec["iteration"] == 0 and cur:
            levels.append(cur); cur = []
        cur.append(rec)
    levels.append(cur)
    sm = s.summary()
    th = s.get_xform_params()
    desc = " | ".join(f"{len(l) - 1} it {sum(x['linear_iterations'] for x in l)} pcg cost {l[-1]['cost']:.9e} acc {sum(int(x['step_is_successful']) for x in l[1:])}" for l in levels)
    print(f"run {r}: scale {np.median(th):.6e} term {sm['termination']} final {sm['final_cost']:.12e}  levels: {desc}", flush=True)
    s.close()
