"""Write a synthetic video as a robust_cvd dataset directory (the on-disk inputs at the drop-in boundary,
SURVEY.md 8b): frames.txt, depth_<model>/depth/frame_%06d.raw (DISPARITY, f32 raw image), flow_list.json,
flow_constraints.dat.  Formats: reference lib/Importer.cpp:197-238, lib/core/CvUtil.cpp:25-36 / 98-107,
flow.py:44-74, lib/FlowConstraints.cpp:191-224.
"""
import json
import os
import struct

import numpy as np

CV_32FC1 = 5


def write_raw_image(path, img):
    """int rows, int cols, int cvType, size_t elemSize, then row-major data (reference lib/core/CvUtil.cpp:98-107).
    [H, W] -> CV_32FC1, [H, W, C] -> CV_32FC(C)."""
    img = np.ascontiguousarray(img, dtype=np.float32)
    assert img.ndim in (2, 3)
    ch = 1 if img.ndim == 2 else img.shape[2]
    with open(path, "wb") as f:
        f.write(struct.pack("<iiiQ", img.shape[0], img.shape[1], CV_32FC1 + 8 * (ch - 1), 4 * ch))
        f.write(img.tobytes())


def write_flow_inputs(base_dir, pairs, flows, masks, colors, dynamic_masks=None):
    """The image inputs of FlowConstraintsCollection::compute (reference lib/FlowConstraints.cpp:226-286, 401-420):
    flow/flow_%06d_%06d.raw (2 x f32, pixels), flow_mask/mask_%06d_%06d.png, color_down/frame_%06d.raw (BGR f32),
    optionally dynamic_mask/frame_%06d.png.  PNGs are written with Pillow."""
    from PIL import Image
    for d in ("flow", "flow_mask", "color_down"):
        os.makedirs(os.path.join(base_dir, d), exist_ok=True)
    for (a, b), fl, mk in zip(np.asarray(pairs).tolist(), flows, masks):
        write_raw_image(os.path.join(base_dir, "flow", f"flow_{a:06d}_{b:06d}.raw"), fl)
        Image.fromarray(np.ascontiguousarray(mk, dtype=np.uint8), "L").save(
            os.path.join(base_dir, "flow_mask", f"mask_{a:06d}_{b:06d}.png"))
    for i, c in enumerate(colors):
        write_raw_image(os.path.join(base_dir, "color_down", f"frame_{i:06d}.raw"), c)
    if dynamic_masks is not None:
        os.makedirs(os.path.join(base_dir, "dynamic_mask"), exist_ok=True)
        for i, m in enumerate(dynamic_masks):
            Image.fromarray(np.ascontiguousarray(m, dtype=np.uint8), "L").save(
                os.path.join(base_dir, "dynamic_mask", f"frame_{i:06d}.png"))


def read_flow_constraints(path, num_pairs, num_triplets):
    """Inverse of write_flow_constraints: (match_separation, {(a, b): [n, 4] f32}, {centre: [n, 6] f32})."""
    with open(path, "rb") as f:
        magic, version, sep = struct.unpack("<IIi", f.read(12))
        assert magic == 0xDEADBEEF and version == 3
        pairs, trips = {}, {}
        for _ in range(num_pairs):
            a, b, n = struct.unpack("<iiQ", f.read(16))
            pairs[(a, b)] = np.frombuffer(f.read(16 * n), dtype=np.float32).reshape(n, 4)
        for _ in range(num_triplets):
            t, n = struct.unpack("<iQ", f.read(12))
            trips[t] = np.frombuffer(f.read(24 * n), dtype=np.float32).reshape(n, 6)
        assert struct.unpack("<I", f.read(4))[0] == 0xDEADBEEF
    return sep, pairs, trips


def write_flow_constraints(path, pairs, offsets, loc, match_separation=10, triplet_centers=()):
    """magic, version 3, matchSeparation, per pair {2 x i32 key, u64 n, n x 4 f32}, per triplet {i32, u64 n (=0)}, magic."""
    with open(path, "wb") as f:
        f.write(struct.pack("<IIi", 0xDEADBEEF, 3, match_separation))
        for p, (a, b) in enumerate(np.asarray(pairs).tolist()):
            n = int(offsets[p + 1] - offsets[p])
            f.write(struct.pack("<iiQ", a, b, n))
            f.write(np.ascontiguousarray(loc[offsets[p]:offsets[p + 1]], dtype=np.float32).tobytes())
        for t in triplet_centers:
            f.write(struct.pack("<iQ", int(t), 0))
        f.write(struct.pack("<I", 0xDEADBEEF))


def write_dataset(base_dir, video, model_type="midas2", fps=30.0):
    os.makedirs(base_dir, exist_ok=True)
    F, W, H = video.num_frames, video.width, video.height
    with open(os.path.join(base_dir, "frames.txt"), "w") as f:
        f.write(f"{F}\n{W}\n{H}\n")
        for i in range(F):
            f.write(f"{i / fps:.6f}\n")
    ddir = os.path.join(base_dir, f"depth_{model_type}", "depth")
    os.makedirs(ddir, exist_ok=True)
    for i in range(F):
        d = video.depth[i]
        disp = np.where(d > 0, 1.0 / np.maximum(d, 1e-30), 0.0).astype(np.float32)  # streams store disparity
        write_raw_image(os.path.join(ddir, f"frame_{i:06d}.raw"), disp)
    # flow_list.json: row 0 is a header (reference flow.py:53, skipped by lib/FlowConstraints.cpp:59)
    with open(os.path.join(base_dir, "flow_list.json"), "w") as f:
        json.dump([["src", "dst"]] + np.asarray(video.pairs).tolist(), f)
    # the reference creates a triplet entry for every interior frame of the range (lib/FlowConstraints.cpp:71-80)
    write_flow_constraints(os.path.join(base_dir, "flow_constraints.dat"), video.pairs, video.offsets, video.loc,
                           triplet_centers=range(1, F - 1))
    for d in ("color_full", "color_down"):
        os.makedirs(os.path.join(base_dir, d), exist_ok=True)
    return base_dir


def read_video_dat(path):
    """Reader of `video.dat` as DepthVideo::save writes it (reference lib/DepthVideo.cpp:300-385, file format 13; the
    reference's own load() cannot read this release's files): frames' pts, the colour streams'
    descriptions and, per depth stream, the transform descriptors and per frame {intrinsics, extrinsics, enabled, depth
    transform parameters, spatial transform parameters}.  Little endian, strings = u64 length + bytes, transforms =
    {i32 XformType, descriptor string} + raw doubles (lib/DepthMapTransform.cpp:270-279, 1493-1506).  This is how results of a
    real reference install (tools/make_reference_golden.py) are read back for comparison."""
    with open(path, "rb") as f:
        raw = f.read()
    pos = 0

    def take(fmt):
        nonlocal pos
        v = struct.unpack_from("<" + fmt, raw, pos)
        pos += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def take_str():
        nonlocal pos
        n = take("Q")
        s = raw[pos:pos + n].decode()
        pos += n
        return s

    def desc_num_params(xtype, text):
        """Number of doubles following a transform descriptor (Xform::params_ size)."""
        name, _, args = text.partition("(")
        args = [a.strip() for a in args.rstrip(")").split(",") if a.strip()]
        nval = {"Scale": 1, "ScaleShift": 2}
        if xtype == 0:  # depth
            if name == "Identity":
                return 0
            if name == "Global":
                return nval[args[0]]
            if name == "Grid":  # Grid(value, Linear|Cubic, x, y, z[, min, max])
                return nval[args[0]] * int(args[2]) * int(args[3]) * int(args[4])
        else:
            if name == "Identity":
                return 0
            if name == "VerticalLinear":
                return 4
            if name == "CornersBilinear":
                return 8
            if name in ("BilinearGrid", "BicubicGrid"):
                return int(args[0]) * int(args[1]) * 2
        raise ValueError(f"unknown transform descriptor {text!r}")

    def take_desc():
        xtype = take("i")
        return xtype, take_str()

    def take_xform():
        xtype, text = take_desc()
        n = desc_num_params(xtype, text)
        vals = np.frombuffer(raw, dtype="<f8", count=n, offset=pos_ref()).copy()
        advance(8 * n)
        return {"desc": text, "params": vals}

    def pos_ref():
        return pos

    def advance(n):
        nonlocal pos
        pos += n

    magic, version, dp_format, num_frames = take("IIIi")
    if magic != 0xDEADBEEF or version < 9:
        raise ValueError("not a video.dat file")
    pts = np.frombuffer(raw, dtype="<f4", count=num_frames, offset=pos).copy()
    pos += 4 * num_frames
    color = []
    for _ in range(take("i")):
        name, dirn, ext = take_str(), take_str(), take_str()
        cvtype, w, h = take("iii")
        gop = take("B")
        if gop:
            raise ValueError("GOP tables are not supported")
        color.append({"name": name, "dir": dirn, "extension": ext, "type": cvtype, "width": w, "height": h})
    depth_streams = []
    for _ in range(take("i")):
        name, dirn = take_str(), take_str()
        ddesc, sdesc = take_desc(), take_desc()
        w, h = take("ii")
        if take("B"):
            raise ValueError("GOP tables are not supported")
        frames = []
        for _f in range(num_frames):
            projection, vfov, hfov, clat, clon = take("iffff")
            position = np.array(take("fff"), np.float32)
            orientation = np.array(take("ffff"), np.float32)  # x, y, z, w
            enabled = bool(take("B"))
            dx, sx = take_xform(), take_xform()
            frames.append({"vfov": vfov, "hfov": hfov, "projection": projection, "position": position,
                           "orientation": orientation, "enabled": enabled, "depth_xform": dx, "spatial_xform": sx})
        depth_streams.append({"name": name, "dir": dirn, "depth_desc": ddesc[1], "spatial_desc": sdesc[1], "width": w,
                              "height": h, "frames": frames})
    duration, w, h, aspect, inv_aspect, magic2 = take("fiiffI")
    if magic2 != 0xDEADBEEF or pos != len(raw):
        raise ValueError("video.dat: trailing marker mismatch")
    return {"version": version, "pts": pts, "color_streams": color, "depth_streams": depth_streams, "duration": duration,
            "width": w, "height": h, "aspect": aspect, "inv_aspect": inv_aspect}


def poses_from_video_dat(video, stream=-1):
    """(position [F, 3], orientation xyzw [F, 4], vfov [F], depth-transform parameters [F, n]) of one depth stream."""
    fr = video["depth_streams"][stream]["frames"]
    return (np.stack([f["position"] for f in fr]), np.stack([f["orientation"] for f in fr]),
            np.array([f["vfov"] for f in fr], np.float32), np.stack([f["depth_xform"]["params"] for f in fr]))
