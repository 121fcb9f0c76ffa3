"""CPU pins of the oracle's restatements added for the SURVEY 8(f) rows: an independent numpy / pure-Python statement of
the same reference lines must reproduce the C++ oracle exactly (the GPU tests then compare the HIP path with the oracle).
  * constraint sampling: reference lib/FlowConstraints.cpp:296-465 (pairs)
  * dense maps: reference lib/DepthMapTransform.cpp:394-449, 950-994"""
import numpy as np

from oracle import oracle as O
from robust_cvd_amd import synth
from robust_cvd_amd.ctypes_types import SpatialXformType, XformDesc


def _greedy_python(W, H, inv_aspect, corner, flow, mask, sep, dyn, dyn_a, dyn_b, min_dyn):
    """Direct transcription of compute(PairKey) + sampleConstraints with float32 arithmetic."""
    f32 = np.float32
    cands = []
    if dyn is not None:
        dh, dw = dyn_a.shape
        sx, sy = f32(dw) / f32(W), f32(dh) / f32(H)
    for iy0 in range(H):
        for ix0 in range(W):
            if not mask[iy0, ix0]:
                continue
            if dyn is not None:
                iy0s = min(int(f32(iy0) * sy + f32(0.5)), dh - 1)
                ix0s = min(int(f32(ix0) * sx + f32(0.5)), dw - 1)
                if not dyn_a[iy0s, ix0s] > min_dyn:
                    continue
            fx1 = f32(ix0) + flow[iy0, ix0, 0]
            fy1 = f32(iy0) + flow[iy0, ix0, 1]
            ix1, iy1 = int(fx1 + f32(0.5)), int(fy1 + f32(0.5))      # C truncation toward zero
            if not (0 <= ix1 < W and 0 <= iy1 < H):
                continue
            if dyn is not None:
                ix1s = min(max(int(fx1 * sx + f32(0.5)), 0), dw - 1)
                iy1s = min(max(int(fy1 * sy + f32(0.5)), 0), dh - 1)
                if not dyn_b[iy1s, ix1s] > min_dyn:
                    continue
            cands.append((corner[iy0, ix0], ix0, iy0, fx1, fy1))
    cands.sort(key=lambda c: -c[0])                                   # Python's sort is stable: ties keep pixel order
    invalid = np.zeros((H, W), bool)
    sxo, syo = f32(1.0) / f32(W), f32(inv_aspect) / f32(H)
    out = []
    for _, ix0, iy0, fx1, fy1 in cands:
        if invalid[iy0, ix0]:
            continue
        out.append((f32(ix0) * sxo, f32(iy0) * syo, fx1 * sxo, fy1 * syo))
        for my in range(max(0, iy0 - sep), min(H - 1, iy0 + sep) + 1):
            for mx in range(max(0, ix0 - sep), min(W - 1, ix0 + sep) + 1):
                if (mx - ix0) ** 2 + (my - iy0) ** 2 <= sep * sep:
                    invalid[my, mx] = True
    return np.asarray(out, dtype=np.float32).reshape(-1, 4)


def test_oracle_pair_sampling_matches_python_transcription():
    F, W, H, P, sep = 3, 40, 24, 4, 4
    rng = np.random.default_rng(9)
    pairs = np.array([[0, 1], [1, 0], [2, 0], [1, 2]], dtype=np.int32)
    corner = (np.round(rng.uniform(0, 1, (F, H, W)) * 16) / 16).astype(np.float32)
    flow = rng.normal(0, 2.0, (P, H, W, 2)).astype(np.float32)
    mask = (rng.uniform(size=(P, H, W)) > 0.2).astype(np.uint8)
    dyn = rng.uniform(0, 10, (F, 13, 21)).astype(np.float32)
    o = O.Oracle()
    o.set_video(F, W, H)
    for dd in (None, dyn):
        off, loc = o.sample_pair_constraints(pairs, corner, flow, mask, sep, dyn_dist=dd, min_dynamic_distance=2.5)
        for p in range(P):
            a, b = pairs[p]
            want = _greedy_python(W, H, o.inv_aspect, corner[a], flow[p], mask[p], sep, dd,
                                  None if dd is None else dd[a], None if dd is None else dd[b], np.float32(2.5))
            got = loc[off[p]:off[p + 1]]
            assert got.shape == want.shape and np.array_equal(got, want), p


def test_oracle_dense_maps_match_gather_hook():
    F, W, H = 2, 40, 24
    v = synth.make_video(F, W, H, seed=3, max_pairs=2)
    dd = XformDesc.grid_depth(5, 4)
    sd = XformDesc.spatial(SpatialXformType.BilinearGrid, 3, 2)
    o = O.Oracle()
    synth.load_into(o, v)
    o.reset_depth_xforms(dd)
    o.reset_spatial_xforms(sd)
    rng = np.random.default_rng(2)
    th = 0.5 + rng.uniform(0, 1, o.get_xform_params(False).shape)
    ph = 0.05 * rng.standard_normal(o.get_xform_params(True).shape)
    o.set_xform_params(th, False)
    o.set_xform_params(ph, True)
    ap, pm, wp = o.apply_depth_xforms(0, F), o.depth_param_maps(0, F), o.spatial_warp_maps(H, W, 0, F)
    xs, ys = np.float32(2.0) / np.float32(W - 1.0), np.float32(2.0) / np.float32(H - 1.0)
    for f in range(F):
        for y in range(0, H, 3):
            for x in range(0, W, 5):
                lx = np.float32(-1.0) + np.float32(x) * xs        # pixel-centre convention, f32, no FMA
                ly = np.float32(1.0) - np.float32(y) * ys
                d = float(v.depth[f, y, x])
                idx, w = O.gather(dd, d, float(lx), float(ly))
                assert np.float32(sum(d * th[f, i] * wi for i, wi in zip(idx, w))) == ap[f, y, x]
                assert sum(th[f, i] * wi for i, wi in zip(idx, w)) == pm[f, y, x]
                idx, w = O.gather(sd, 0.0, float(lx), float(ly))
                assert np.float32(sum(ph[f, 2 * i] * wi for i, wi in zip(idx, w))) == wp[f, y, x, 0]
                assert np.float32(sum(ph[f, 2 * i + 1] * wi for i, wi in zip(idx, w))) == wp[f, y, x, 1]
