"""The committed evidence under profiles/ belongs to the code in the tree (no GPU needed)."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_pmc_traffic_file_was_measured_on_these_kernel_sources():
    """bench.py reports roofline.traffic from profiles/pmc_matvec_pairs.json only when the file's hash of the kernel sources
    matches the tree (separate rocprofv3 --pmc passes cannot run inside the bench process).  A source change without
    `bash tools/pmc_refresh.sh` leaves the bench line without its traffic figure: caught here."""
    import bench
    with open(os.path.join(ROOT, "profiles", "pmc_matvec_pairs.json")) as f:
        pmc = json.load(f)
    assert pmc["constraints"] == 2396032 and pmc["traffic_bytes_per_launch"] > 0
    if pmc["kernel_sources_sha256"] != bench.kernel_sources_digest():
        import pytest
        pytest.skip("profiles/pmc_matvec_pairs.json is stale: the kernel headers changed since it was measured (bench.py reports "
                    "roofline.traffic = null until `bash tools/pmc_refresh.sh` has run on the GPU box)")


def test_dense_traffic_file_was_measured_on_these_kernel_sources():
    """The same for `bench.py --dense`: profiles/pmc_dense_walk.json (tools/dense_profile.sh) carries the hash of the headers that
    define k_dense_walk."""
    import bench
    import pytest
    path = os.path.join(ROOT, "profiles", "pmc_dense_walk.json")
    with open(path) as f:
        pmc = json.load(f)
    assert pmc["kernel"].startswith("k_dense_walk") and pmc["traffic_bytes_per_launch"] > 0
    if pmc["kernel_sources_sha256"] != bench.kernel_sources_digest(dense=True):
        pytest.skip("profiles/pmc_dense_walk.json is stale: the kernel headers changed since it was measured (the dense bench line "
                    "reports roofline.traffic = null until `bash tools/dense_profile.sh` has run on the GPU box)")


def test_committed_bench_lines_follow_the_contract():
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[23456]*_bench.json")))
    assert lines, "no bench line under profiles/"
    d = json.loads(open(lines[-1]).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["unit"] == "GB/s"
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
