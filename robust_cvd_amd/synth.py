"""Seeded synthetic inputs for the geometric-consistency optimizer (bench.py + tests).

Produces exactly what the reference's optimizer consumes:
  * per-frame source depth maps (what DepthFrame::sourceDepth() returns after the disparity->depth
    inversion of reference lib/DepthStream.cpp:193-216),
  * the directed frame-pair list of `flow_list.json` (re-statement of the reference's `hierarchical2`
    sampler, reference utils/frame_sampling.py:77-120, two_way=True as in reference video.py:181-183),
  * per-pair flow constraints in the reference's `[0,1] x [0,invAspect]` convention
    (reference lib/FlowConstraints.cpp:352-397, 446-465: loc0 = integer source pixel * (1/w, invAspect/h),
    loc1 = (pixel + flow) * (1/w, invAspect/h)).

Scene: the inside of a box room (convex => no occlusions, every ray hits a wall), cameras on a smooth
random walk.  Camera model = the optimizer's own (reference lib/PoseOptimizer.cpp:174-221):
  X = t + D * R * (ndc.x * fx, ndc.y * fy, -1),   fx = fy * aspect.
The "network" depth handed to the optimizer is the true depth divided by a per-frame scale and a smooth
multiplicative field, so that a per-frame scale + a grid deformation can undo it.

Everything here is numpy; nothing in this file is part of the optimizer arithmetic.
"""
from dataclasses import dataclass, field

import numpy as np


# ----------------------------------------------------------------------------------------------------
# frame pairs
# ----------------------------------------------------------------------------------------------------
def hierarchical_pairs(num_frames, two_way=True, min_dist=1, max_dist=None, include_mid_point=True,
                       extra_offsets=False):
    """Re-statement of SamplePairs.sample_hierarchical(2) (reference utils/frame_sampling.py:77-120).

    extra_offsets = k > 0 densifies the start frames of the long-range levels: level l steps by 2^max(0, l - k)
    instead of 2^(l - 1) ("add levels with half-step offsets until P ~ 4000").  At 300 frames:
    k = 1 is the reference's own list (1766 directed pairs), 2 -> 2332, 3 (= True) -> 2874, 4 -> 3374, 5 -> 3808,
    6 -> 4140 = the "~4k pairs" flow list BASELINE.json's north_star names.
    """
    if max_dist is None:
        max_dist = num_frames - 1
    min_level = int(np.ceil(np.log2(min_dist)))
    max_level = int(np.floor(np.log2(max_dist)))
    signs = (-1, 1) if two_way else (1,)
    pairs = set()
    for level in range(min_level, max_level + 1):
        dist = 1 << level
        step_level = max(0, level - 1) if include_mid_point else level
        if extra_offsets:
            step_level = max(0, level - (3 if extra_offsets is True else int(extra_offsets)))
        step = 1 << step_level
        for start in range(0, num_frames, step):
            for sign in signs:
                end = start + sign * dist
                if end < 0 or end >= num_frames:
                    continue
                pairs.add((start, end))
    return sorted(pairs)


# ----------------------------------------------------------------------------------------------------
# scene + cameras
# ----------------------------------------------------------------------------------------------------
def rodrigues(w):
    """Rotation matrices [..., 3, 3] of angle-axis vectors [..., 3]."""
    w = np.asarray(w, dtype=np.float64)
    th = np.linalg.norm(w, axis=-1, keepdims=True)
    k = np.divide(w, th, out=np.zeros_like(w), where=th > 0)
    K = np.zeros(w.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2] = -k[..., 2], k[..., 1]
    K[..., 1, 0], K[..., 1, 2] = k[..., 2], -k[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -k[..., 1], k[..., 0]
    s = np.sin(th)[..., None]
    c = np.cos(th)[..., None]
    eye = np.broadcast_to(np.eye(3), K.shape)
    return eye + s * K + (1.0 - c) * (K @ K)


ROOM = np.array([4.0, 3.0, 6.0])  # half extents of the box room (x, y, z)


def scene_depth(t, R, fx, fy, nx, ny):
    """Depth D (along the optical axis) of the ray through ndc (nx, ny): first wall hit of t + D*R*c."""
    c = np.stack([nx * fx, ny * fy, -np.ones_like(nx)], axis=-1)  # [..., 3]
    d = np.einsum("...ij,...j->...i", R, c)
    best = np.full(nx.shape, np.inf)
    for axis in range(3):
        for sgn in (-1.0, 1.0):
            denom = d[..., axis]
            num = sgn * ROOM[axis] - t[..., axis]
            with np.errstate(divide="ignore", invalid="ignore"):
                s = num / denom
            s = np.where((denom * sgn > 0) & (s > 0), s, np.inf)
            best = np.minimum(best, s)
    return best


def smooth_walk(rng, n, sigma, taps=9):
    steps = rng.normal(0.0, sigma, size=(n, 3))
    taps = max(1, min(taps, n))
    kernel = np.ones(taps) / taps
    for k in range(3):
        steps[:, k] = np.convolve(steps[:, k], kernel, mode="same")
    walk = np.cumsum(steps, axis=0)
    return walk - walk[0]


@dataclass
class SyntheticVideo:
    num_frames: int
    width: int
    height: int
    aspect: float
    inv_aspect: float
    depth: np.ndarray           # [F, H, W] float32 source depth handed to the optimizer
    true_depth: np.ndarray      # [F, H, W] float32
    true_t: np.ndarray          # [F, 3]
    true_w: np.ndarray          # [F, 3] angle-axis
    true_fy: float
    frame_scale: np.ndarray     # [F]  s_f: depth = true_depth / (s_f * (1 + amp * field))
    pairs: np.ndarray           # [P, 2] int32 directed pairs (sorted like the reference's std::map)
    offsets: np.ndarray         # [P + 1] int64
    loc: np.ndarray             # [C, 4] float32 (loc0.xy, loc1.xy)
    is_static: np.ndarray       # [C] uint8
    meta: dict = field(default_factory=dict)

    @property
    def num_constraints(self):
        return int(self.offsets[-1])


def _field(rng, H, W, gh=3, gw=4):
    """Smooth multiplicative field in [-1, 1]: bilinear upsampling of a tiny random grid."""
    g = rng.uniform(-1.0, 1.0, size=(gh, gw))
    ys = np.linspace(0, gh - 1, H)
    xs = np.linspace(0, gw - 1, W)
    y0 = np.clip(np.floor(ys).astype(int), 0, gh - 2)
    x0 = np.clip(np.floor(xs).astype(int), 0, gw - 2)
    ry = (ys - y0)[:, None]
    rx = (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return (1 - ry) * ((1 - rx) * a + rx * b) + ry * ((1 - rx) * c + rx * d)


def make_video(num_frames, width, height, seed=1234, pairs=None, spacing=12.5, flow_noise_px=0.25,
               field_amp=0.10, trans_sigma=0.02, rot_sigma_deg=0.5, focal_long=0.3461538376301239,
               scale_range=(0.5, 2.0), dense=False, max_pairs=None, extra_offsets=False, outlier_fraction=0.0,
               outlier_px=20.0):
    """Build a SyntheticVideo. `spacing` ~ 12.5 px reproduces the density of the reference's greedy disk
    sampling with matchSeparation = 10 (about 590 constraints per pair at 384x224). dense=True emits every
    in-bounds pixel (the reference's matchSeparation = 0 regime)."""
    rng = np.random.default_rng(seed)
    F, W, H = int(num_frames), int(width), int(height)
    aspect = np.float32(W) / np.float32(H)
    inv_aspect = np.float32(1.0) / aspect
    A = float(aspect)
    fy = focal_long / A if A >= 1.0 else focal_long  # reference lib/PoseOptimizer.cpp:1155-1157
    fx = fy * A

    t = smooth_walk(rng, F, trans_sigma)
    w = smooth_walk(rng, F, np.deg2rad(rot_sigma_deg))
    R = rodrigues(w)

    # depth maps, pixel-edge NDC convention of the constraints
    xs = -1.0 + 2.0 * np.arange(W) / W
    ys = 1.0 - 2.0 * np.arange(H) / H
    nx, ny = np.meshgrid(xs, ys)
    true_depth = np.empty((F, H, W), dtype=np.float32)
    depth = np.empty((F, H, W), dtype=np.float32)
    frame_scale = np.exp(rng.uniform(np.log(scale_range[0]), np.log(scale_range[1]), size=F))
    for f in range(F):
        D = scene_depth(t[f], R[f], fx, fy, nx, ny)
        true_depth[f] = D
        depth[f] = D / (frame_scale[f] * (1.0 + field_amp * _field(rng, H, W)))

    if pairs is None:
        pairs = hierarchical_pairs(F, two_way=True, extra_offsets=extra_offsets)
    pairs = np.asarray(sorted(map(tuple, pairs)), dtype=np.int32).reshape(-1, 2)
    if max_pairs is not None:
        pairs = pairs[:max_pairs]
    P = pairs.shape[0]

    # candidate source pixels per pair
    if dense:
        gx, gy = np.meshgrid(np.arange(W), np.arange(H))
        base = np.stack([gx.ravel(), gy.ravel()], axis=-1).astype(np.float64)  # [M, 2]
        M = base.shape[0]
    else:
        row_h = spacing * np.sqrt(3.0) / 2.0
        rows = int(np.floor(H / row_h)) + 2
        cols = int(np.floor(W / spacing)) + 2
        jj, ii = np.meshgrid(np.arange(cols), np.arange(rows))
        bx = (jj + 0.5 * (ii % 2)) * spacing
        by = ii * row_h
        base = np.stack([bx.ravel(), by.ravel()], axis=-1)
        M = base.shape[0]

    loc_chunks, off = [], [0]
    scale_x = np.float32(1.0) / np.float32(W)
    scale_y = inv_aspect / np.float32(H)
    chunk = max(1, int(4_000_000 // max(M, 1)))
    for p0 in range(0, P, chunk):
        pp = pairs[p0:p0 + chunk]
        n = pp.shape[0]
        if dense:
            pix = np.broadcast_to(base[None], (n, M, 2))
        else:
            shift = rng.uniform(-spacing, 0.0, size=(n, 1, 2))
            jitter = rng.uniform(-0.9, 0.9, size=(n, M, 2))
            pix = base[None] + shift + jitter
        ix = np.rint(pix[..., 0])
        iy = np.rint(pix[..., 1])
        inb = (ix >= 0) & (ix < W) & (iy >= 0) & (iy < H)
        a, b = pp[:, 0], pp[:, 1]
        nxa = -1.0 + 2.0 * ix / W
        nya = 1.0 - 2.0 * iy / H
        Da = scene_depth(t[a][:, None, :], R[a][:, None, :, :], fx, fy, nxa, nya)
        ca = np.stack([nxa * fx, nya * fy, -np.ones_like(nxa)], axis=-1)
        X = t[a][:, None, :] + Da[..., None] * np.einsum("pij,pmj->pmi", R[a], ca)
        q = np.einsum("pji,pmj->pmi", R[b], X - t[b][:, None, :])  # R_b^T (X - t_b)
        z = -q[..., 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            u = q[..., 0] / z / fx
            v = q[..., 1] / z / fy
        x1 = (u + 1.0) * 0.5 * W + rng.normal(0.0, flow_noise_px, size=u.shape)
        y1 = (1.0 - v) * 0.5 * H + rng.normal(0.0, flow_noise_px, size=v.shape)
        if outlier_fraction > 0.0:  # gross flow errors (what the robust loss is for); no draw at the default 0: seeds keep their videos
            bad = rng.uniform(size=u.shape) < outlier_fraction
            x1 = x1 + bad * rng.uniform(-outlier_px, outlier_px, size=u.shape)
            y1 = y1 + bad * rng.uniform(-outlier_px, outlier_px, size=u.shape)
        # reference lib/FlowConstraints.cpp:446-449: rounded target pixel must be in bounds
        fx1 = x1.astype(np.float32)
        fy1 = y1.astype(np.float32)
        ix1 = np.floor(fx1 + np.float32(0.5))
        iy1 = np.floor(fy1 + np.float32(0.5))
        ok = inb & (z > 1e-3) & np.isfinite(x1) & np.isfinite(y1) & (ix1 >= 0) & (ix1 < W) & (iy1 >= 0) & (iy1 < H)
        for k in range(n):
            sel = ok[k]
            l = np.empty((int(sel.sum()), 4), dtype=np.float32)
            l[:, 0] = ix[k][sel].astype(np.float32) * scale_x
            l[:, 1] = iy[k][sel].astype(np.float32) * scale_y
            l[:, 2] = fx1[k][sel] * scale_x
            l[:, 3] = fy1[k][sel] * scale_y
            loc_chunks.append(l)
            off.append(off[-1] + l.shape[0])
    loc = np.concatenate(loc_chunks, axis=0) if loc_chunks else np.zeros((0, 4), np.float32)
    offsets = np.asarray(off, dtype=np.int64)
    return SyntheticVideo(
        num_frames=F, width=W, height=H, aspect=float(aspect), inv_aspect=float(inv_aspect),
        depth=depth, true_depth=true_depth, true_t=t, true_w=w, true_fy=fy, frame_scale=frame_scale,
        pairs=pairs, offsets=offsets, loc=loc, is_static=np.ones(loc.shape[0], dtype=np.uint8),
        meta={"seed": seed, "spacing": spacing, "flow_noise_px": flow_noise_px, "dense": dense,
              "field_amp": field_amp})


def make_dense_flows(video: SyntheticVideo, flow_noise_px=0.25, seed=99, invalid_fraction=0.02):
    """Flow and mask IMAGES of every directed pair of `video` (what flow/flow_%06d_%06d.raw and flow_mask/mask_*.png hold,
    reference lib/FlowConstraints.cpp:226-255): flow [P, H, W, 2] float32 in pixels from the true geometry + noise,
    mask [P, H, W] uint8 (0 where the target falls outside the image or behind the camera, plus a random
    `invalid_fraction` standing in for the forward/backward consistency check).  Input of the dense mode."""
    rng = np.random.default_rng(seed)
    F, W, H = video.num_frames, video.width, video.height
    A = float(video.aspect)
    fy = video.true_fy
    fx = fy * A
    R = rodrigues(video.true_w)
    t = video.true_t
    gx, gy = np.meshgrid(np.arange(W), np.arange(H))
    nx = -1.0 + 2.0 * gx / W
    ny = 1.0 - 2.0 * gy / H
    P = len(video.pairs)
    flow = np.zeros((P, H, W, 2), np.float32)
    mask = np.zeros((P, H, W), np.uint8)
    cam = np.stack([nx * fx, ny * fy, -np.ones_like(nx)], axis=-1)
    for k, (a, b) in enumerate(np.asarray(video.pairs).tolist()):
        D = scene_depth(t[a], R[a], fx, fy, nx, ny)
        X = t[a] + D[..., None] * (cam @ R[a].T)
        q = (X - t[b]) @ R[b]
        z = -q[..., 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            u = q[..., 0] / z / fx
            v = q[..., 1] / z / fy
        x1 = (u + 1.0) * 0.5 * W + rng.normal(0.0, flow_noise_px, size=u.shape)
        y1 = (1.0 - v) * 0.5 * H + rng.normal(0.0, flow_noise_px, size=v.shape)
        ok = (z > 1e-3) & np.isfinite(x1) & np.isfinite(y1) & (x1 > -0.5) & (x1 < W - 0.5) & (y1 > -0.5) & (y1 < H - 0.5)
        ok &= rng.uniform(size=ok.shape) >= invalid_fraction
        flow[k, ..., 0] = np.where(ok, x1 - gx, 0.0)
        flow[k, ..., 1] = np.where(ok, y1 - gy, 0.0)
        mask[k] = np.where(ok, 255, 0)
    return flow, mask


def dense_constraints_from_flows(video: SyntheticVideo, flow, mask):
    """The constraint list the reference's FlowConstraintsCollection::compute produces from these images with
    matchSeparation = 0 (lib/FlowConstraints.cpp:436-460, 352-397: every masked pixel whose target int(x + flow + 0.5) is
    in bounds; loc = pixel * (1 / w, invAspect / h) in float).  Returns (offsets [P + 1] int64, loc [C, 4] float32) in
    row-major pixel order per pair (the reference orders by corner response: irrelevant to the cost, a sum)."""
    W, H = video.width, video.height
    sx = np.float32(1.0) / np.float32(W)
    sy = np.float32(video.inv_aspect) / np.float32(H)
    gx, gy = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    off, chunks = [0], []
    for k in range(flow.shape[0]):
        fx1 = gx + flow[k, ..., 0].astype(np.float32)
        fy1 = gy + flow[k, ..., 1].astype(np.float32)
        with np.errstate(invalid="ignore"):
            ix1 = np.trunc(fx1 + np.float32(0.5))   # C++ float -> int conversion truncates towards zero
            iy1 = np.trunc(fy1 + np.float32(0.5))
        sel = (mask[k] != 0) & np.isfinite(fx1) & np.isfinite(fy1) & (ix1 >= 0) & (ix1 < W) & (iy1 >= 0) & (iy1 < H)
        l = np.empty((int(sel.sum()), 4), np.float32)
        l[:, 0] = gx[sel] * sx
        l[:, 1] = gy[sel] * sy
        l[:, 2] = fx1[sel] * sx
        l[:, 3] = fy1[sel] * sy
        chunks.append(l)
        off.append(off[-1] + l.shape[0])
    return np.asarray(off, np.int64), (np.concatenate(chunks, 0) if chunks else np.zeros((0, 4), np.float32))


def make_triplets(video: SyntheticVideo, spacing=25.0, seed=77, flow_noise_px=0.25, dynamic_fraction=0.25):
    """Triplet constraints of the scene-flow smoothness loss (reference lib/FlowConstraints.h:116-205, keyed by the
    centre frame): points sampled in frame c, projected with the true geometry into c-1 and c+1 (+ flow noise).
    Returns (centers [T] int32, offsets [T+1] int64, loc6 [C, 6] float32 = loc(c-1), loc(c), loc(c+1), is_static [C])."""
    rng = np.random.default_rng(seed)
    F, W, H = video.num_frames, video.width, video.height
    A = float(video.aspect)
    fy = video.true_fy
    fx = fy * A
    R = rodrigues(video.true_w)
    t = video.true_t
    scale_x = np.float32(1.0) / np.float32(W)
    scale_y = np.float32(video.inv_aspect) / np.float32(H)
    row_h = spacing * np.sqrt(3.0) / 2.0
    rows = int(np.floor(H / row_h)) + 2
    cols = int(np.floor(W / spacing)) + 2
    jj, ii = np.meshgrid(np.arange(cols), np.arange(rows))
    base = np.stack([((jj + 0.5 * (ii % 2)) * spacing).ravel(), (ii * row_h).ravel()], axis=-1)
    centers, off, chunks, flags = [], [0], [], []
    for c in range(1, F - 1):
        pix = base + rng.uniform(-spacing, 0.0, size=(1, 2)) + rng.uniform(-0.9, 0.9, size=base.shape)
        ix, iy = np.rint(pix[:, 0]), np.rint(pix[:, 1])
        ok = (ix >= 0) & (ix < W) & (iy >= 0) & (iy < H)
        nx = -1.0 + 2.0 * ix / W
        ny = 1.0 - 2.0 * iy / H
        D = scene_depth(t[c], R[c], fx, fy, nx, ny)
        cam = np.stack([nx * fx, ny * fy, -np.ones_like(nx)], axis=-1)
        X = t[c] + D[:, None] * (cam @ R[c].T)
        l = np.empty((base.shape[0], 6), dtype=np.float32)
        l[:, 2] = ix.astype(np.float32) * scale_x
        l[:, 3] = iy.astype(np.float32) * scale_y
        for k, o in ((0, c - 1), (4, c + 1)):
            q = (X - t[o]) @ R[o]  # R_o^T (X - t_o)
            z = -q[:, 2]
            with np.errstate(divide="ignore", invalid="ignore"):
                u = q[:, 0] / z / fx
                v = q[:, 1] / z / fy
            x1 = ((u + 1.0) * 0.5 * W + rng.normal(0.0, flow_noise_px, size=u.shape)).astype(np.float32)
            y1 = ((1.0 - v) * 0.5 * H + rng.normal(0.0, flow_noise_px, size=v.shape)).astype(np.float32)
            ok &= (z > 1e-3) & np.isfinite(x1) & np.isfinite(y1)
            ok &= (np.floor(x1 + np.float32(0.5)) >= 0) & (np.floor(x1 + np.float32(0.5)) < W)
            ok &= (np.floor(y1 + np.float32(0.5)) >= 0) & (np.floor(y1 + np.float32(0.5)) < H)
            l[:, k] = x1 * scale_x
            l[:, k + 1] = y1 * scale_y
        l = l[ok]
        centers.append(c)
        chunks.append(l)
        flags.append((rng.uniform(size=l.shape[0]) >= dynamic_fraction).astype(np.uint8))
        off.append(off[-1] + l.shape[0])
    loc6 = np.concatenate(chunks, axis=0) if chunks else np.zeros((0, 6), np.float32)
    st = np.concatenate(flags) if flags else np.zeros((0,), np.uint8)
    return np.asarray(centers, np.int32), np.asarray(off, np.int64), loc6, st


def load_into(binding, video: SyntheticVideo, focal_long=0.3461538376301239):
    """Upload a SyntheticVideo through the common C-ABI surface (product Solver or test Oracle)."""
    binding.set_video(video.num_frames, video.width, video.height, video.aspect, video.inv_aspect)
    binding.set_depth_all(video.depth)
    binding.set_pair_constraints(video.pairs, video.offsets, video.loc, video.is_static)
    binding.reset_poses(focal_long)


# ----------------------------------------------------------------------------------------------------
# gauge-invariant comparison helpers (tests / bench): the optimizer fixes neither the global rigid
# transform nor (exactly) the global scale, so poses are compared through relative quantities.
# ----------------------------------------------------------------------------------------------------
def quat_to_matrix(q_xyzw):
    q = np.asarray(q_xyzw, dtype=np.float64)
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z)
    R[..., 0, 1] = 2 * (x * y - z * w)
    R[..., 0, 2] = 2 * (x * z + y * w)
    R[..., 1, 0] = 2 * (x * y + z * w)
    R[..., 1, 1] = 1 - 2 * (x * x + z * z)
    R[..., 1, 2] = 2 * (y * z - x * w)
    R[..., 2, 0] = 2 * (x * z - y * w)
    R[..., 2, 1] = 2 * (y * z + x * w)
    R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def align_similarity(src, dst):
    """Umeyama: s, R, t minimising |s R src + t - dst|."""
    src = np.asarray(src, dtype=np.float64)
    dst = np.asarray(dst, dtype=np.float64)
    mu_s, mu_d = src.mean(0), dst.mean(0)
    xs, xd = src - mu_s, dst - mu_d
    cov = xd.T @ xs / src.shape[0]
    U, S, Vt = np.linalg.svd(cov)
    d = np.sign(np.linalg.det(U) * np.linalg.det(Vt))
    D = np.diag([1.0, 1.0, d])
    Rm = U @ D @ Vt
    var = (xs ** 2).sum() / src.shape[0]
    s = np.trace(np.diag(S) @ D) / var if var > 0 else 1.0
    tt = mu_d - s * Rm @ mu_s
    return s, Rm, tt


def relative_pose_error(pos_a, quat_a, pos_b, quat_b):
    """Gauge-aligned relative pose error between two solutions of the same problem:
    positions after a similarity alignment (relative to the trajectory extent) and the largest
    relative-rotation angle (radians) after removing the common rotation."""
    pos_a = np.asarray(pos_a, np.float64)
    pos_b = np.asarray(pos_b, np.float64)
    s, Rm, tt = align_similarity(pos_a, pos_b)
    extent = max(np.linalg.norm(pos_b - pos_b.mean(0), axis=1).max(), 1e-12)
    pos_err = np.linalg.norm((s * (Rm @ pos_a.T).T + tt) - pos_b, axis=1).max() / extent
    Ra = quat_to_matrix(quat_a)
    Rb = quat_to_matrix(quat_b)
    rel_a = np.einsum("ij,fjk->fik", Ra[0].T, Ra)
    rel_b = np.einsum("ij,fjk->fik", Rb[0].T, Rb)
    dR = np.einsum("fji,fjk->fik", rel_a, rel_b)
    ang = np.arccos(np.clip((np.trace(dR, axis1=1, axis2=2) - 1.0) * 0.5, -1.0, 1.0))
    return float(pos_err), float(ang.max())
