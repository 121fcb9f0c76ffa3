"""Development aid (GPU): constraint sampling at the benchmark's size (300 frames, 1766 directed pairs, 384 x 224)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robust_cvd_amd import api, synth
from oracle.oracle import Oracle

F, W, H, SEP = 300, 384, 224, 10
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1766
rng = np.random.default_rng(0)
pairs = np.asarray(synth.hierarchical_pairs(F), dtype=np.int32)[:P]
corner = rng.uniform(0, 1, (F, H, W)).astype(np.float32)
flow = rng.normal(0, 2.0, (P, H, W, 2)).astype(np.float32)
mask = (rng.uniform(size=(P, H, W)) > 0.1).astype(np.uint8)
s = api.Solver(0); s.set_video(F, W, H)
s.sample_pair_constraints(pairs[:8], corner, flow[:8], mask[:8], SEP)      # warm-up
t0 = time.perf_counter(); off, loc = s.sample_pair_constraints(pairs, corner, flow, mask, SEP); dt = time.perf_counter() - t0
print(f"GPU: {P} pairs, {off[-1]} constraints ({off[-1] / P:.0f} per pair) in {dt * 1e3:.1f} ms incl. host<->device copies "
      f"({(flow.nbytes + mask.nbytes + corner.nbytes) / 1e9:.2f} GB in)")
o = Oracle(); o.set_video(F, W, H)
n = 16
t0 = time.perf_counter(); off2, loc2 = o.sample_pair_constraints(pairs[:n], corner, flow[:n], mask[:n], SEP); dc = time.perf_counter() - t0
assert np.array_equal(off2, off[:n + 1]) and np.array_equal(loc2, loc[:off[n]])
print(f"CPU oracle (1 thread): {n} pairs in {dc * 1e3:.1f} ms = {dc / n * 1e3:.2f} ms per pair -> {dc / n * P:.2f} s for {P} pairs")
