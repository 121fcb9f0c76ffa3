#!/usr/bin/env python3
"""Development aid (GPU): where a workgroup of the hot product k_matvec_pairs_fast spends its life.
Library variant with wall-clock stamps (100 MHz):  python -c "from robust_cvd_amd import build; build.build_variant('mvprof', ['CVD_MV_PROFILE'], ['cvd_matvec'])"
then  python tools/mv_profile.py [pairs_level]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
from robust_cvd_amd import api, synth
api.load_library(variant="mvprof")  # (before bench / Solver load the product library)
import bench
from robust_cvd_amd.ctypes_types import OptParams

level = int(sys.argv[1]) if len(sys.argv) > 1 else 6
v = synth.make_video(300, 384, 224, seed=bench.SEED, extra_offsets=level)
s = api.Solver(0)
p = OptParams.defaults()
bench.prepare(s, v, p)
s.set_options(pcg_lockstep=1)
p.max_iterations = 2
s.pose_optimization_step(p, p.depth_deform_reg_final, convert_poses=False)   # the stamps of the LAST product launch stay
lib = api.load_library()
buf = (C.c_ulonglong * (4096 * 8))()
assert lib.cvd_debug_mv_profile(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
us = lambda x: x / 100.0
end = a[:, 3]
print("workgroups", len(a), " launch span us %.1f" % us(end.max() - t0))
st = us(a[:, 0] - t0)
print("start offsets us: 25/50/75/100 %%: %s" % np.percentile(st, [25, 50, 75, 100]).round(1))
loop_end = a[:, 4:8].max(1)
loop_first = a[:, 4:8].min(1)
for name, x in (("prologue (loads, p, E)", us(a[:, 1] - a[:, 0])), ("loop, slowest wave", us(loop_end - a[:, 1])),
                ("loop, fastest wave", us(loop_first - a[:, 1])), ("reduction", us(a[:, 2] - loop_end)), ("write-out", us(a[:, 3] - a[:, 2])),
                ("total", us(a[:, 3] - a[:, 0]))):
    print(f"{name:26s} min {x.min():6.2f}  median {np.median(x):6.2f}  mean {x.mean():6.2f}  max {x.max():6.2f}")
tot = us(a[:, 3] - a[:, 0])
print("sum of workgroup lifetimes / (256 CUs x 4 slots: four waves per SIMD since round 5) = %.1f us" % (tot.sum() / 1024))
# occupancy over time: how many workgroups are alive
ts = np.linspace(0, us(end.max() - t0), 30)
alive = [(int(((st <= t) & (us(end - t0) > t)).sum())) for t in ts]
print("alive workgroups over the launch:", alive)
