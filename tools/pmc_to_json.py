#!/usr/bin/env python3
"""profiles/pmc_matvec_pairs.json from the two rocprofv3 --pmc passes of tools/profile_round.sh (FETCH_SIZE, WRITE_SIZE).

usage: pmc_to_json.py <round dir with pmc_fetch/ pmc_write/ bench.json> <tag>
The JSON records the SHA-256 of the kernel sources it was measured on (bench.kernel_sources_digest) and the workload's
constraint count: bench.py only reports `roofline.traffic` from it when both match what it is benchmarking.
Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in KiB-like units of
1024 B; on gfx950 FETCH_SIZE counts a wide coalesced 16 B/lane stream at half its bytes, so it is doubled; WRITE_SIZE is
taken as is."""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def avg(d, counter, kernel):
    tot, n = 0.0, 0
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(fn)):
            if row["Counter_Name"] == counter and kernel in row["Kernel_Name"]:
                tot += float(row["Counter_Value"]); n += 1
    return tot / max(n, 1), n


root, tag = sys.argv[1], sys.argv[2]
kernel = "k_matvec_pairs_fast<4"  # (any workgroup-size instantiation)
f, nf = avg(os.path.join(root, "pmc_fetch"), "FETCH_SIZE", kernel)
w, nw = avg(os.path.join(root, "pmc_write"), "WRITE_SIZE", kernel)
bench_line = json.loads([l for l in open(os.path.join(root, "bench.json")) if l.startswith("{")][-1])
out = {
    "kernel": kernel,
    "kernel_sources_sha256": bench.kernel_sources_digest(),
    "constraints": bench_line["config"]["constraints"], "pairs": bench_line["config"]["pairs"],
    "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --kernel-include-regex "
               "k_matvec_pairs_fast --output-format csv -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary "
               "--pcg-lockstep (lockstep: no early-exit launches in the average)",
    "fetch_size_kb_per_launch_raw": f, "write_size_kb_per_launch_raw": w, "dispatches": nf,
    "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE counts a wide coalesced 16 B/lane stream at "
                  "half its bytes -> doubled; WRITE_SIZE uncalibrated, taken as is",
    "traffic_bytes_per_launch": (2.0 * f + w) * 1024.0,
    "source_files": [f"profiles/{tag}_pmc_FETCH_SIZE.csv", f"profiles/{tag}_pmc_WRITE_SIZE.csv"],
}
print(json.dumps(out, indent=1))
