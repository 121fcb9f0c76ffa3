"""Synthetic-input generator: pair sampler re-statement and constraint conventions."""
import numpy as np

from robust_cvd_amd import synth


def test_hierarchical2_pair_counts():
    # counts computed from the reference sampler (utils/frame_sampling.py:77-120, two_way=True): SURVEY.md 2
    assert [len(synth.hierarchical_pairs(n)) for n in (30, 100, 300, 1000)] == [156, 572, 1766, 5958]


def test_hierarchical2_structure():
    pairs = synth.hierarchical_pairs(40)
    s = set(pairs)
    assert all((b, a) in s for a, b in pairs)            # two-way
    assert all(0 <= a < 40 and 0 <= b < 40 and a != b for a, b in pairs)
    for a, b in pairs:
        d = abs(a - b)
        assert d & (d - 1) == 0                          # power-of-two distances
        lvl = int(np.log2(d))
        assert min(a, b) % (1 << max(0, lvl - 1)) == 0 or max(a, b) % (1 << max(0, lvl - 1)) == 0
    assert pairs == sorted(pairs)                        # std::map<pair<int,int>> order


def test_video_conventions():
    v = synth.make_video(6, 96, 56, seed=11)
    assert v.depth.shape == (6, 56, 96) and v.depth.dtype == np.float32
    assert v.loc.dtype == np.float32 and v.loc.shape[1] == 4
    assert (v.depth > 0).all()
    # loc0 is an integer pixel * (1/w, invAspect/h); loc in [0,1] x [0,invAspect]
    px = v.loc[:, 0] * 96
    py = v.loc[:, 1] / np.float32(v.inv_aspect) * 56
    assert np.abs(px - np.rint(px)).max() < 1e-3 and np.abs(py - np.rint(py)).max() < 1e-3
    assert v.loc[:, 0].min() >= 0 and v.loc[:, 0].max() < 1
    assert v.loc[:, 1].min() >= 0 and v.loc[:, 1].max() < v.inv_aspect
    assert v.offsets[0] == 0 and v.offsets[-1] == v.loc.shape[0] and (np.diff(v.offsets) > 0).all()
    # deterministic
    v2 = synth.make_video(6, 96, 56, seed=11)
    assert np.array_equal(v.loc, v2.loc) and np.array_equal(v.depth, v2.depth)


def test_zero_noise_flow_is_geometrically_consistent():
    """With zero noise, reprojecting a source pixel with the TRUE poses/depth lands on loc1."""
    v = synth.make_video(5, 64, 40, seed=2, flow_noise_px=0.0, field_amp=0.0, scale_range=(1.0, 1.0))
    R = synth.rodrigues(v.true_w)
    A = v.aspect
    fy, fx = v.true_fy, v.true_fy * A
    p = 3
    a, b = v.pairs[p]
    sl = slice(v.offsets[p], v.offsets[p + 1])
    loc = v.loc[sl].astype(np.float64)
    nx, ny = -1 + 2 * loc[:, 0], 1 - 2 * loc[:, 1] / v.inv_aspect
    ix = (loc[:, 0] * 64 + 1e-6).astype(int)
    iy = (loc[:, 1] / v.inv_aspect * 40 + 1e-6).astype(int)
    D = v.true_depth[a][iy, ix].astype(np.float64)
    c = np.stack([nx * fx, ny * fy, -np.ones_like(nx)], -1)
    X = v.true_t[a] + D[:, None] * (c @ R[a].T)
    q = (X - v.true_t[b]) @ R[b]
    u, w = q[:, 0] / -q[:, 2] / fx, q[:, 1] / -q[:, 2] / fy
    nx1, ny1 = -1 + 2 * loc[:, 2], 1 - 2 * loc[:, 3] / v.inv_aspect
    assert np.abs(u - nx1).max() < 1e-4 and np.abs(w - ny1).max() < 1e-4


def test_pairs_match_reference_sampler():
    """Pinned against the REAL reference: vectors minted by importing reference utils/frame_sampling.py
    (tests/golden/reference_py/make_pairs_golden.py), every mode that runs through sample_hierarchical."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_py", "reference_pairs.npz"))
    checked = 0
    for n in (2, 3, 5, 8, 17, 30, 100, 300):
        for tw in (True, False):
            got = np.asarray(synth.hierarchical_pairs(n, two_way=tw), dtype=np.int32).reshape(-1, 2)
            assert np.array_equal(got, g[f"h2_n{n}_tw{int(tw)}"]), (n, tw)
        got = np.asarray(synth.hierarchical_pairs(n, max_dist=1, include_mid_point=False), dtype=np.int32).reshape(-1, 2)
        assert np.array_equal(got, g[f"consecutive_n{n}"]), n
        got = np.asarray(synth.hierarchical_pairs(n, min_dist=2, max_dist=min(9, max(2, n - 1)), include_mid_point=False),
                         dtype=np.int32).reshape(-1, 2)
        assert np.array_equal(got, g[f"h1_n{n}_min2_max9"]), n
        checked += 4
    assert checked == 32
