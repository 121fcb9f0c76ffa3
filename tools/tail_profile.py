#!/usr/bin/env python3
"""Development aid (GPU): where the workgroups of k_pcg_tail (finish + update of a PCG iteration in one launch) spend their life.
Library variant with wall-clock stamps (100 MHz):
    python -c "from robust_cvd_amd import build; build.build_variant('tailprof', ['CVD_TAIL_PROFILE'], ['cvd_matvec'])"
then  python tools/tail_profile.py [pairs_level]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
from robust_cvd_amd import api, synth
api.load_library(variant="tailprof")  # (before bench / Solver load the product library)
import bench
from robust_cvd_amd.ctypes_types import OptParams

level = int(sys.argv[1]) if len(sys.argv) > 1 else 6
v = synth.make_video(300, 384, 224, seed=bench.SEED, extra_offsets=level)
s = api.Solver(0)
p = OptParams.defaults()
bench.prepare(s, v, p)
s.set_options(pcg_lockstep=1)
p.max_iterations = 2
s.pose_optimization_step(p, p.depth_deform_reg_final, convert_poses=False)   # the stamps of the LAST tail launch stay
lib = api.load_library()
buf = (C.c_ulonglong * (1024 * 8))()
assert lib.cvd_debug_tail_profile(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
a = a[a[:, 0] > 0]
F = v.num_frames
t0 = a[:, 0].min()
us = lambda x: x / 100.0
print("workgroups", len(a), " launch span us %.1f" % us(a[:, 4].max() - t0))
# workgroups behind the frames': the exact dense level's (two frames each; none with the temporal pose level), then the temporal
# levels' (45 hats of the depth-grid level, 8 modes of the temporal pose level)
nExtra = len(a) - F
nD = (F + 1) // 2 if nExtra >= (F + 1) // 2 else 0
groups = [("frame workgroups", a[:F]), ("dense-level workgroups", a[F:F + nD])]
rest = a[F + nD:]
dbg = s.temporal_debug()
nTl = dbg["S"] * max(1, (dbg["nn"] - 1 + 9) // 10) if dbg else 0   # (hats x node ranges of 10 intervals: TlStep::parts)
groups += [("depth-grid level's workgroups", rest[:nTl]), ("temporal pose level's workgroups", rest[nTl:])]
for name, rows in groups:
    if not len(rows):
        continue
    print(name, len(rows))
    for label, x in (("start offset", us(rows[:, 0] - t0)), ("finish half", us(rows[:, 1] - rows[:, 0])),
                     ("operand requests", us(rows[:, 2] - rows[:, 1])), ("grid barrier + alpha", us(rows[:, 3] - rows[:, 2])),
                     ("update half", us(rows[:, 4] - rows[:, 3])), ("barrier release at", us(rows[:, 3] - t0)),
                     ("end at", us(rows[:, 4] - t0))):
        print(f"  {label:22s} min {x.min():6.2f}  median {np.median(x):6.2f}  mean {x.mean():6.2f}  max {x.max():6.2f}")
    if "level's" in name and rows[:, 5].min() > 0:   # (tlLevelRows' own stamps)
        for label, x in (("  node sums (q_T)", us(rows[:, 5] - rows[:, 3])), ("  rows + update", us(rows[:, 6] - rows[:, 5])),
                         ("  interpolation + share", us(rows[:, 7] - rows[:, 6])), ("  ticket", us(rows[:, 4] - rows[:, 7]))):
            print(f"  {label:22s} min {x.min():6.2f}  median {np.median(x):6.2f}  mean {x.mean():6.2f}  max {x.max():6.2f}")
