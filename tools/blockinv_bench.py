"""How the per-frame block inverse (k_block_inverse_mfma) scales with the number of frames at B = 177: one workgroup per frame, 256 CUs.
usage (GPU box): rocprofv3 --kernel-trace --stats -d out -- python tools/blockinv_bench.py   (kernel durations in the trace)
or: python tools/blockinv_bench.py  (wall clock around the debug entry point: includes the transfers -- differences only)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robust_cvd_amd import api  # noqa: E402

B = 177
rng = np.random.default_rng(0)
s = api.Solver(0)
for n in (128, 256, 300, 512, 600):
    a = rng.standard_normal((n, B, B))
    blocks = a @ a.transpose(0, 2, 1) + B * np.eye(B)
    s.block_inverse_debug(blocks)
    t0 = time.perf_counter()
    out, fl = s.block_inverse_debug(blocks)
    print(n, "frames:", round((time.perf_counter() - t0) * 1e3, 2), "ms wall (with transfers), failed pivots", fl, flush=True)
s.close()
