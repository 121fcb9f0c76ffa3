#!/usr/bin/env python3
"""Fold the margin logs of repeated GPU suite runs (tests/margins.py) into one table: per assertion the worst value / limit over all
runs and its run-to-run spread.  usage: margin_report.py <log> [<log> ...]   (exit 1 when a tolerance uses more than 1/3 of its limit)"""
import collections
import json
import sys

rows = collections.defaultdict(list)
for path in sys.argv[1:]:
    for line in open(path):
        r = json.loads(line)
        rows[(r["test"], r["name"], r.get("kind", "tolerance"))].append((r["value"], r["limit"]))
bad = 0
print(f"# {len(sys.argv) - 1} run(s), {len(rows)} logged assertions; policy: tolerances -- worst value / limit <= 0.333; iteration-count "
      f"comparisons (count: |a - b| against 10 - 20 % + slack; ratio: measured ratio at least 5 % inside its limit) -- must hold")
LIMIT = {"tolerance": 1.0 / 3.0, "count": 1.0, "ratio": 0.95}
for (test, name, kind), vals in sorted(rows.items(), key=lambda kv: -max(v / l if l else 0.0 for v, l in kv[1])):
    worst = max(v / l if l else 0.0 for v, l in vals)
    vs = [v for v, _ in vals]
    over = worst > LIMIT[kind] + 1e-12
    flag = f"  <-- over {LIMIT[kind]:.2f}" if over else ""
    bad += over
    print(f"{worst:7.3f}  {kind:9s} n={len(vals):2d}  min {min(vs):.3e}  max {max(vs):.3e}  limit {vals[0][1]:.3e}  {test} :: {name}{flag}")
sys.exit(1 if bad else 0)
