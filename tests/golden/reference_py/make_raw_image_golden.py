#!/usr/bin/env python3
"""Mint tests/golden/reference_py/raw_image_golden.npz: the BYTES the reference's own writer of the raw float image format
(utils/image_io.py:138-173 `save_raw_float32_image`, the Python twin of the C++ `fwriteimg` / `freadimg` that DepthStream and the
flow reader use) produces for small one- and two-channel images, with the arrays they encode.  Build container only
(needs /root/reference):

    python tests/golden/reference_py/make_raw_image_golden.py
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from tests import reference_residuals as rres  # noqa: E402  (the cv2 stub + sys.path of the reference checkout)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "raw_image_golden.npz")


def main():
    rres._reference_modules()
    from utils import image_io
    rng = np.random.default_rng(77)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for name, shape in (("depth_5x7", (5, 7)), ("flow_4x6x2", (4, 6, 2)), ("color_3x5x3", (3, 5, 3))):
            img = rng.standard_normal(shape).astype(np.float32)
            path = os.path.join(d, name + ".raw")
            image_io.save_raw_float32_image(path, img)
            back = image_io.load_raw_float32_image(path)
            assert np.array_equal(np.asarray(back).reshape(shape), img)
            out[name + "/image"] = img
            out[name + "/bytes"] = np.frombuffer(open(path, "rb").read(), np.uint8)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
