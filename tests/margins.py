"""Assertion-margin policy of the GPU suite (Next #1c).

Solves on the device are not bit-reproducible run to run (LDS / global f64 atomics accumulate in arrival order), and the LM / PCG
stopping rules amplify the last bits into +-1 PCG iteration and ~1e-4 along near-gauge directions.  Every numeric assertion on a
solve's OUTCOME therefore goes through `below` / `close_count` / `same_count`, which
  * assert, and
  * when CVD_MARGIN_LOG names a file, append {test, name, value, limit} to it.
tools/margin_report.py folds the logs of repeated suite runs (tools/gpu_round_check.sh <tag> 3) into profiles/r05_margins.log; the
policy: the worst value / limit over all runs is <= 1/3 for tolerances, iteration-count comparisons carry >= 10 % (+ 2) slack,
LM-iteration counts of two paths may differ by one (a stopping test decided in the 7th digit of the cost change).
"""
import json
import os


def _log(name, value, limit, kind="tolerance"):
    path = os.environ.get("CVD_MARGIN_LOG")
    if not path:
        return
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    with open(path, "a") as f:
        f.write(json.dumps({"test": test, "name": name, "value": float(value), "limit": float(limit), "kind": kind}) + "\n")


def deterministic_build():
    """The suite runs against lib/libcvd_hip_det.so (pytest --lib-variant det): every accumulation in index order, a solve repeats
    bit for bit, so comparisons of two paths hold their TIGHT limits there (the relaxed limits of the product build --
    atomics in arrival order -- would let a real divergence at the 1e-5 level pass)."""
    from robust_cvd_amd import api
    return api.loaded_variant() == "det"


def limit(product, det):
    """The limit of an assertion on the product build / on the deterministic build."""
    return det if deterministic_build() else product


def below(name, value, limit, info=None, kind="tolerance"):
    """value < limit: logged with its margin.  kind "tolerance": the 1/3 policy applies; "ratio": a ratio of two iteration counts
    (>= 5 % between the measured value and the limit: the limit must still catch a lost benefit)."""
    _log(name, value, limit, kind)
    assert value < limit, (name, value, limit, info)


def close_count(name, a, b, rel=0.1, slack=2):
    """Two iteration counts agree to rel (>= 10 %) + slack."""
    lim = rel * max(a, b) + slack
    _log(name, abs(a - b), lim, "count")
    assert abs(a - b) <= lim, (name, a, b)


def same_count(name, a, b, slack=1):
    """LM-iteration counts of two solver paths: equal up to one iteration (a stopping test decided in the last digits); equal on
    the deterministic build."""
    if deterministic_build():
        slack = 0
    _log(name, abs(a - b), slack + 1, "count")
    assert abs(a - b) <= slack, (name, a, b)
