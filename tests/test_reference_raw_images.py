"""Reference-held pin of the RAW FLOAT IMAGE format: depth_<model>/depth/frame_%06d.raw, flow/flow_*.raw and
color_down/frame_*.raw are written by the reference's Python (utils/image_io.py:138-173 `save_raw_float32_image`) and read by its C++
(`freadimg`, lib/core/CvUtil.cpp:98-107; DepthStream, the flow reader of FlowConstraintsCollection).  Here:

  * always: robust_cvd_amd.dataset_io.write_raw_image produces, byte for byte, what the reference's writer produced for the committed
    images (tests/golden/reference_py/raw_image_golden.npz, minted by make_raw_image_golden.py next to it), and the drop-in
    lib_python's DepthStream reads a frame file made of those reference bytes;
  * with /root/reference mounted: the live writer against dataset_io's, the live reader on dataset_io's files, and a whole depth
    stream written by the reference's function and imported through lib_python.
"""
import importlib
import os
import sys

import numpy as np
import pytest

from robust_cvd_amd import build as _b
from robust_cvd_amd import dataset_io, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_py", "raw_image_golden.npz")
CASES = ("depth_5x7", "flow_4x6x2", "color_3x5x3")


@pytest.fixture(scope="module")
def lib():
    d = os.path.dirname(_b.build_lib_python())
    if d not in sys.path:
        sys.path.insert(0, d)
    return importlib.import_module("lib_python")


@pytest.mark.parametrize("name", CASES)
def test_writer_is_byte_identical_to_the_references(name, tmp_path):
    g = np.load(GOLDEN)
    path = str(tmp_path / "img.raw")
    dataset_io.write_raw_image(path, g[name + "/image"])
    assert open(path, "rb").read() == g[name + "/bytes"].tobytes()


def test_depth_stream_reads_a_frame_the_reference_wrote(lib, tmp_path):
    """A depth stream whose frame files are reference bytes: DepthStream (lib_python) must hand back 1 / disparity of exactly those
    values (the stream stores disparity: loaders/video_dataset.py, lib/DepthStream.cpp)."""
    g = np.load(GOLDEN)
    disp = np.abs(g["depth_5x7/image"]) + 0.5          # (positive disparities)
    v = synth.make_video(3, 7, 5, seed=88, spacing=3)
    base = dataset_io.write_dataset(str(tmp_path / "video"), v)
    # overwrite frame 1 with a file in the REFERENCE's bytes: header of the golden + our payload would not be a pin, so the payload is
    # the golden's own image (made positive by rewriting through the byte layout the golden documents: header 20 bytes, row-major f32)
    raw = bytearray(g["depth_5x7/bytes"].tobytes())
    assert len(raw) == 20 + disp.size * 4
    raw[20:] = disp.astype("<f4").tobytes()
    with open(os.path.join(base, "depth_midas2", "depth", "frame_000001.raw"), "wb") as f:
        f.write(bytes(raw))
    dv = lib.DepthVideo()
    lib.DepthVideoImporter.importVideo(dv, base, True)
    dv.createDepthStream("depth_midas2", "depth_midas2", [v.width, v.height])
    src = dv.depthStream(0).frame(1).sourceDepth()
    np.testing.assert_allclose(src, 1.0 / disp, rtol=2e-7)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference checkout not mounted")
def test_live_reference_writer_and_reader(lib, tmp_path):
    from tests import reference_residuals as rres
    rres._reference_modules()
    from utils import image_io
    rng = np.random.default_rng(5)
    for shape in ((9, 13), (6, 8, 2), (4, 4, 3)):
        img = rng.standard_normal(shape).astype(np.float32)
        a, b = str(tmp_path / "ref.raw"), str(tmp_path / "ours.raw")
        image_io.save_raw_float32_image(a, img)
        dataset_io.write_raw_image(b, img)
        assert open(a, "rb").read() == open(b, "rb").read()
        assert np.array_equal(np.asarray(image_io.load_raw_float32_image(b)).reshape(shape), img)
    # a whole depth stream written by the reference's function, imported by the drop-in module
    v = synth.make_video(4, 48, 28, seed=89, spacing=8)
    base = dataset_io.write_dataset(str(tmp_path / "video"), v)
    for i in range(v.num_frames):
        image_io.save_raw_float32_image(os.path.join(base, "depth_midas2", "depth", f"frame_{i:06d}.raw"), 1.0 / v.depth[i])
    dv = lib.DepthVideo()
    lib.DepthVideoImporter.importVideo(dv, base, True)
    dv.createDepthStream("depth_midas2", "depth_midas2", [v.width, v.height])
    for i in range(v.num_frames):
        np.testing.assert_allclose(dv.depthStream(0).frame(i).sourceDepth(), v.depth[i], rtol=3e-7)
