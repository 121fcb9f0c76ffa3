"""Shared synthetic inputs for the flow-guided filter tests (seeded; arithmetic parity, not a plausible scene)."""
import numpy as np


def make_case(n, w, h, dw=None, dh=None, seed=0, flow_sigma=1.2, mask_keep=0.9):
    rng = np.random.default_rng(seed)
    dw, dh = dw or w, dh or h
    depth = rng.uniform(1.0, 6.0, (n, dh, dw)).astype(np.float32)
    cams = np.zeros((n, 9), np.float32)
    cams[:, :3] = rng.normal(0, 0.05, (n, 3))
    q = np.concatenate([rng.normal(0, 0.03, (n, 3)), np.ones((n, 1))], axis=1)
    cams[:, 3:7] = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    cams[:, 7] = 0.9 + rng.uniform(-0.02, 0.02, n)       # hFov
    cams[:, 8] = 0.55 + rng.uniform(-0.02, 0.02, n)      # vFov
    links = max(n - 1, 0)
    ff = rng.normal(0, flow_sigma, (links, h, w, 2)).astype(np.float32)
    fb = rng.normal(0, flow_sigma, (links, h, w, 2)).astype(np.float32)
    mf = (rng.uniform(size=(links, h, w)) < mask_keep).astype(np.uint8) * 255
    mb = (rng.uniform(size=(links, h, w)) < mask_keep).astype(np.uint8) * 255
    return dict(depth=depth, cameras=cams, flow_fwd=ff, mask_fwd=mf, flow_bwd=fb, mask_bwd=mb, inv_aspect=np.float32(h / w))
