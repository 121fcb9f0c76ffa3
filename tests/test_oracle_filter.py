"""CPU pin of the oracle's flowGuidedFilter restatement: an independent numpy-float32 transcription of reference
lib/Processor.cpp:428-585 + lib/DepthVideo.cpp:637-681 must reproduce the C++ oracle (mean: to float rounding of
exp; median: exactly, since it returns one of the sample depths)."""
import numpy as np
import pytest

from oracle.oracle import Oracle
from tests.filter_cases import make_case

f32 = np.float32


def _rot(q, v):
    qv, w = q[:3], q[3]
    uv = np.cross(qv, v).astype(f32)
    uv = (uv + uv).astype(f32)
    return (v + w * uv + np.cross(qv, uv).astype(f32)).astype(f32)


def filter_python(c, frame_radius, spatial_radius, median, first, count):
    depth, cams = c["depth"], c["cameras"]
    n, dh, dw = depth.shape
    ia = f32(c["inv_aspect"])
    h, w = (c["flow_fwd"].shape[1:3]) if n > 1 else (dh, dw)
    cam = []
    for k in range(n):
        q = cams[k, 3:7]
        cam.append(dict(pos=cams[k, :3], right=_rot(q, np.array([1, 0, 0], f32)), up=_rot(q, np.array([0, 1, 0], f32)),
                        front=_rot(q, np.array([0, 0, -1], f32)), th=f32(np.tan(cams[k, 7] / f32(2))), tv=f32(np.tan(cams[k, 8] / f32(2)))))
    out = np.zeros((count, h, w), f32)
    for o in range(count):
        fr = first + o
        ref = cam[fr]
        f0, f1 = max(0, fr - frame_radius), min(n - 1, fr + frame_radius)

        def sample(lx, ly, fi):
            nx, ny = f32(lx / f32(w)), f32(f32(ly / f32(h)) * ia)
            x = max(0, min(dw - 1, int(f32(nx * f32(dw)) + f32(0.5))))
            y = max(0, min(dh - 1, int(f32(f32(ny / ia) * f32(dh)) + f32(0.5))))
            d = depth[fi, y, x]
            cc = cam[fi]
            a = f32(f32(f32(-1) + f32(2) * nx) * cc["th"])
            b = f32(f32(f32(1) - f32(f32(2) * ny) / ia) * cc["tv"])
            ray = ((cc["front"] + cc["right"] * a).astype(f32) + cc["up"] * b).astype(f32)
            p = (cc["pos"] + ray * d).astype(f32)
            t = ((p - ref["pos"]).astype(f32) * ref["front"]).astype(f32)
            return f32(f32(t[0] + t[1]) + t[2])

        for y in range(h):
            for x in range(w):
                S = []
                refd = None
                for wy in range(max(0, y - spatial_radius), min(h - 1, y + spatial_radius) + 1):
                    for wx in range(max(0, x - spatial_radius), min(w - 1, x + spatial_radius) + 1):
                        S.append(sample(f32(wx), f32(wy), fr))
                        if wx == x and wy == y:
                            refd = S[-1]
                        for step, stop, flow, mask, off in ((1, f1, c["flow_fwd"], c["mask_fwd"], -1), (-1, f0, c["flow_bwd"], c["mask_bwd"], 0)):
                            lx, ly = f32(wx), f32(wy)
                            fi = fr + step
                            while (fi <= stop) if step > 0 else (fi >= stop):
                                e = fi + off
                                ix, iy = min(int(lx + f32(0.5)), w - 1), min(int(ly + f32(0.5)), h - 1)
                                if not mask[e, iy, ix]:
                                    break
                                lx, ly = f32(lx + flow[e, iy, ix, 0]), f32(ly + flow[e, iy, ix, 1])
                                ix, iy = int(lx + f32(0.5)), int(ly + f32(0.5))
                                if ix < 0 or ix >= w or iy < 0 or iy >= h:
                                    break
                                S.append(sample(lx, ly, fi))
                                fi += step
                W = [f32(np.exp(f32(-(max(s, refd) / min(s, refd)) * f32(3)))) for s in S]
                if median:
                    ws = f32(0)
                    for v in W:
                        ws = f32(ws + v)
                    half = f32(ws / f32(2))
                    cum = f32(0)
                    for i in np.argsort(np.array(S, f32), kind="stable"):
                        cum = f32(cum + W[i])
                        if cum >= half:
                            out[o, y, x] = S[i]
                            break
                else:
                    ds, ws = f32(0), f32(0)
                    for s, v in zip(S, W):
                        ds, ws = f32(ds + f32(s * v)), f32(ws + v)
                    out[o, y, x] = f32(ds / ws) if ws > 0 else 0
    return out


@pytest.mark.parametrize("median,sr,fr", [(False, 0, 2), (True, 0, 2), (False, 1, 1), (True, 1, 3)])
def test_oracle_filter_matches_python_transcription(median, sr, fr):
    c = make_case(5, 10, 7, seed=3 + sr)
    o = Oracle()
    got = o.flow_guided_filter(c["depth"], c["cameras"], c["flow_fwd"], c["mask_fwd"], c["flow_bwd"], c["mask_bwd"],
                               c["inv_aspect"], fr, spatial_radius=sr, median=median, first=1, count=3)
    ref = filter_python(c, fr, sr, median, 1, 3)
    assert got.shape == ref.shape
    assert np.allclose(got, ref, rtol=2e-6, atol=0)
    if median:   # one of the sample depths: exact unless exp's last bit moves the half-weight crossing
        assert np.mean(got == ref) > 0.98


def test_oracle_filter_radius_zero_is_the_identity_on_own_depth():
    """frameRadius = spatialRadius = 0: one sample per pixel (the pixel's own depth along its own axis)."""
    c = make_case(2, 9, 6, seed=8)
    o = Oracle()
    got = o.flow_guided_filter(c["depth"], c["cameras"], c["flow_fwd"], c["mask_fwd"], c["flow_bwd"], c["mask_bwd"],
                               c["inv_aspect"], 0)
    # ray . front = 1 for a unit quaternion, so the axial depth equals the stored depth up to float rounding
    assert np.allclose(got, c["depth"], rtol=1e-5)
