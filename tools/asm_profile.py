"""Development aid (GPU, library built with -DCVD_ASM_PROFILE): phase timestamps of k_assemble_fast per frame."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robust_cvd_amd import api, synth
from robust_cvd_amd.ctypes_types import *
mode = sys.argv[1] if len(sys.argv) > 1 else "bilinear"
F = 300 if mode == "bilinear" else 100
v = synth.make_video(F, 384, 224, seed=1234)
s = api.Solver(0); synth.load_into(s, v)
p = OptParams.defaults()
s.reset_spatial_xforms(XformDesc.spatial())
s.reset_depth_xforms(XformDesc.grid_depth(17, 10) if mode == "bilinear" else XformDesc.grid_depth(4, 4, cubic=True))
pose = np.zeros((F, 7)); pose[:, 6] = 0.2
for _ in range(3):
    s.evaluate(p, 0.1, None)
lib = api._load() if hasattr(api, "_load") else None
lib = lib or C.CDLL(os.path.join(os.path.dirname(api.__file__), "lib", "libcvd_hip.so"))
buf = (C.c_ulonglong * (2048 * 16))()
assert lib.cvd_debug_asm_profile(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 16).astype(np.int64)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
us = lambda x: x / 100.0  # 100 MHz
end = a.max(1)
print("parts", len(a), "kernel span us", us(end.max() - t0))
st = us(a[:, 0] - t0)
print("start offsets us: median/max", np.median(st), st.max(), " started late (>30us):", int(np.sum(st > 30)))
for name, (i, j) in {"zero": (0, 1), "loop(all waves)": (1, 2)}.items():
    x = us(a[:, j] - a[:, i]); print(f"{name:18s} min {x.min():8.1f} median {np.median(x):8.1f} max {x.max():8.1f}")
x = us(end - a[:, 2]); print(f"{'after loop':18s} min {x.min():8.1f} median {np.median(x):8.1f} max {x.max():8.1f}")
x = us(end - a[:, 0]); print(f"{'total':18s} min {x.min():8.1f} median {np.median(x):8.1f} max {x.max():8.1f}  sum/256 {x.sum()/256:8.1f}")
one = a[(a[:, 3] > 0)]
for name, (i, j) in {"reduce PP": (2, 13), "shared": (13, 14), "regulariser": (14, 15), "cost+combine": (15, 3), "writeout": (3, 12)}.items():
    x = us(one[:, j] - one[:, i]); print(f"{name:18s} min {x.min():8.1f} median {np.median(x):8.1f} max {x.max():8.1f}")
