#!/usr/bin/env python3
"""Mint tests/golden/reference_py/reference_pairs.npz by IMPORTING the reference's own frame-pair sampler.

The one piece of the path's inputs that is Python in the reference: `SamplePairs.sample_hierarchical2`
(reference utils/frame_sampling.py:77-120), which produces the flow_list the optimizer's constraints are built on
(`two_way=True` as in reference video.py:181-183).  This script can only run where /root/reference exists (the
build container); the vectors it writes travel with the repository and pin `robust_cvd_amd.synth.hierarchical_pairs`
against the real reference code (tests/test_synth.py::test_pairs_match_reference_sampler).

    python tests/golden/reference_py/make_pairs_golden.py
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/utils/frame_sampling.py"

sys.path.insert(0, "/root/reference")  # the module imports its sibling utils.frame_range
spec = importlib.util.spec_from_file_location("ref_frame_sampling", REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

out = {}
for n in (2, 3, 5, 8, 17, 30, 100, 300):
    for two_way in (True, False):
        pairs = sorted(tuple(p) for p in ref.SamplePairs.sample_hierarchical2(n, two_way))
        out[f"h2_n{n}_tw{int(two_way)}"] = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)
    # the other modes the sampler offers on the same code path (consecutive = hierarchical with max_dist 1)
    out[f"consecutive_n{n}"] = np.asarray(sorted(tuple(p) for p in ref.SamplePairs.sample_consecutive(n, True)),
                                          dtype=np.int32).reshape(-1, 2)
    out[f"h1_n{n}_min2_max9"] = np.asarray(
        sorted(tuple(p) for p in ref.SamplePairs.sample_hierarchical(n, True, min_dist=2, max_dist=min(9, max(2, n - 1)))),
        dtype=np.int32).reshape(-1, 2)
np.savez_compressed(os.path.join(HERE, "reference_pairs.npz"), **out)
print({k: v.shape[0] for k, v in out.items()})
