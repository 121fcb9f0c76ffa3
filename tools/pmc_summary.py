#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV into per-kernel averages (development / profiles aid).

usage: pmc_summary.py <dir-with-*_counter_collection.csv> [out.csv]
"""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for fn in files:
        with open(fn) as f:
            for row in csv.DictReader(f):
                name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
                name = name.split("(")[0].replace("void ", "")
                c = row.get("Counter_Name") or row.get("Counter Name")
                v = float(row.get("Counter_Value") or row.get("Counter Value") or 0.0)
                a = acc[name][c]
                a[0] += v
                a[1] += 1
    rows = []
    for k, cs in acc.items():
        for c, (s, n) in cs.items():
            rows.append((k, c, n, s / max(n, 1), s))
    rows.sort(key=lambda r: -r[4])
    out = sys.argv[2] if len(sys.argv) > 2 else None
    lines = ["kernel,counter,dispatches,avg_per_dispatch,total"]
    for r in rows:
        lines.append(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]:.6g},{r[4]:.6g}")
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
