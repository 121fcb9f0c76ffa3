"""The deterministic build (cvd_kernels.h: CVD_DETERMINISTIC; robust_cvd_amd.build.build_deterministic) reproduces a solve bit for bit.

The product build accumulates through LDS f64 atomics issued by several waves of a workgroup: two runs of one solve differ in the
last bits and the stopping rules turn that into +-1 PCG iteration (what made round 4's GPU record flaky).  The deterministic build
issues every such accumulation from one wave and folds partial results in index order; this test repeats the final level's solve of
a 4140-pair-style problem (100 frames to keep it short) from the same state in a fresh process and requires IDENTICAL PCG counts,
final-cost bits and pose bits in every repeat.  (The variant library must be the first load of the library in its process: hence
the subprocess.)
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["final_level_solve", "whole_pipeline"])
def test_deterministic_build_repeats_a_solve_bit_for_bit(what):
    from robust_cvd_amd import build as b
    b.build_deterministic()
    extra = ["--pipeline"] if what == "whole_pipeline" else ["--iterations", "6"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "det_check.py"), "--variant", "det", "--frames", "100", "--reps", "4"] + extra,
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    reps = [l.split(" ", 2)[2] for l in out.stdout.splitlines() if l.startswith("rep ")]
    assert len(reps) == 4, out.stdout
    assert len(set(reps)) == 1, "\n".join(reps)
    assert "[robust_cvd_amd] loading the development variant" in out.stderr and "libcvd_hip_det.so" in out.stderr
