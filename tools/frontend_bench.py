"""Development aid (GPU): the kernels either side of the optimizer at the benchmark's size (300 frames, 384 x 224):
corner response, dynamic-mask distance transform, flow-guided filter (the pipeline's radius 4).  Kernel times come from
HIP events around the launches (host<->device copies excluded); GB/s = algorithmic bytes / kernel time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robust_cvd_amd import api
from oracle.oracle import Oracle
from tests.filter_cases import make_case

F, W, H = 300, 384, 224
rng = np.random.default_rng(0)
s = api.Solver(0)
o = Oracle()
px = F * W * H

bgr = rng.uniform(0, 1, (F, H, W, 3)).astype(np.float32)
s.corner_min_eigenval(bgr[:4])
out, ms = s.corner_min_eigenval(bgr, timing=True)
alg = px * (12 + 4)            # BGR in, response out
stream = px * (12 + 4 + 4 + 12 + 12 + 4)   # + gray and cov round trips of the three-kernel chain
print(f"corner response: {F} frames in {ms:.3f} ms kernel time; algorithmic {alg / 1e6:.0f} MB -> {alg / ms / 1e6:.0f} GB/s "
      f"(streamed by the three kernels {stream / 1e6:.0f} MB -> {stream / ms / 1e6:.0f} GB/s)")
n = 8
t0 = time.perf_counter(); ref = o.corner_min_eigenval(bgr[:n]); dc = time.perf_counter() - t0
assert np.array_equal(ref, out[:n])
print(f"  CPU oracle (1 thread): {dc / n * 1e3:.2f} ms per frame -> {dc / n * F:.2f} s for {F} frames")

mask = np.where(rng.uniform(size=(F, H // 2, W // 2)) < 0.002, 0, 255).astype(np.uint8)
for y in range(0, 40):
    mask[:, 30 + y, 50:90] = 0
s.dynamic_distance(mask[:4])
dd, ms = s.dynamic_distance(mask, timing=True)
print(f"distance transform: {F} masks {W // 2}x{H // 2} in {ms:.3f} ms kernel time (one workgroup per mask, {F} in flight)")
t0 = time.perf_counter(); ref = o.dynamic_distance(mask[:n]); dc = time.perf_counter() - t0
assert np.array_equal(ref, dd[:n])
print(f"  CPU oracle (1 thread): {dc / n * 1e3:.2f} ms per mask -> {dc / n * F:.2f} s for {F} masks")

c = make_case(F, W, H, seed=1, flow_sigma=2.0)
args = (c["depth"], c["cameras"], c["flow_fwd"], c["mask_fwd"], c["flow_bwd"], c["mask_bwd"], c["inv_aspect"])
for R, sr, med in ((4, 0, False), (4, 0, True), (2, 1, False)):
    s.flow_guided_filter(c["depth"][:6], c["cameras"][:6], c["flow_fwd"][:5], c["mask_fwd"][:5], c["flow_bwd"][:5], c["mask_bwd"][:5],
                         c["inv_aspect"], R, spatial_radius=sr, median=med)   # warm-up
    out, ms = s.flow_guided_filter(*args, R, spatial_radius=sr, median=med, timing=True)
    # per output pixel: (2 sr + 1)^2 windows x (1 own depth texel + up to 2R chain steps of flow 8 B + mask 1 B + depth 4 B)
    alg = px * ((2 * sr + 1) ** 2 * (4 + 2 * R * 13) + 4)
    print(f"flow guided filter R={R} r={sr} {'median' if med else 'mean'}: {F} frames in {ms:.3f} ms kernel time; "
          f"<= {alg / 1e6:.0f} MB gathered -> {alg / ms / 1e6:.0f} GB/s")
    if not med and sr == 0:
        m = 2
        t0 = time.perf_counter()
        ref = o.flow_guided_filter(c["depth"][:R + m], c["cameras"][:R + m], c["flow_fwd"][:R + m - 1], c["mask_fwd"][:R + m - 1],
                                   c["flow_bwd"][:R + m - 1], c["mask_bwd"][:R + m - 1], c["inv_aspect"], R, first=0, count=m)
        dc = time.perf_counter() - t0
        print(f"  CPU oracle (1 thread): {dc / m * 1e3:.1f} ms per frame -> {dc / m * F:.2f} s for {F} frames")
