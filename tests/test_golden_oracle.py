"""The oracle reproduces the committed golden vectors (tests/golden/make_golden.py) on any host."""
import pytest

from oracle.oracle import Oracle
from tests.helpers import evaluate_golden, golden_cases, load_golden, rel


@pytest.mark.parametrize("name", golden_cases())
def test_oracle_matches_golden(name):
    g = load_golden(name)
    ev = evaluate_golden(Oracle(), g)
    assert ev["num_residual_blocks"] == int(g["num_residual_blocks"])
    # identical code + compiler flags (-ffp-contract=off, no fast-math): libm differences only
    assert abs(ev["cost"] - float(g["cost"])) <= 1e-12 * abs(float(g["cost"]))
    assert rel(ev["gradient"], g["gradient"]) < 1e-11
    assert rel(ev["hdiag"], g["hdiag"]) < 1e-11


def test_golden_cases_cover_the_scope_table():
    names = golden_cases()
    assert len(names) >= 7
    assert any("cubic" in n for n in names) and any("17x10" in n for n in names)
    assert sum("triplets" in n for n in names) >= 2   # scene-flow smoothness, both implemented Laplacians
