#!/usr/bin/env python3
"""profiles/pmc_matvec_pairs.json from the two rocprofv3 --pmc passes of tools/profile_round.sh (FETCH_SIZE, WRITE_SIZE).

usage: pmc_to_json.py <round dir with pmc_fetch/ pmc_write/ bench.json> <tag>      -> profiles/pmc_matvec_pairs.json
       pmc_to_json.py <round dir with pmc_FETCH_SIZE/ pmc_WRITE_SIZE/ bench_dense_cfg.json> <tag> dense   -> profiles/pmc_dense_walk.json
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
dense = len(sys.argv) > 3 and sys.argv[3] == "dense"
if dense:
    # the dense mode's one-walk Jacobian evaluation (tools/dense_profile.sh): its loads are 64 B lines of flow / 32 B of depth / 8 B of
    # mask per run and scattered 4 B depth gathers -- not the 128 B requests the guide's doubling is about: FETCH_SIZE taken as is
    kernel, factor = "k_dense_walk<4", 1.0
    f, nf = avg(os.path.join(root, "pmc_FETCH_SIZE"), "FETCH_SIZE", kernel)
    w, nw = avg(os.path.join(root, "pmc_WRITE_SIZE"), "WRITE_SIZE", kernel)
    bench_file, cmd_kernel, cmd_bench = "bench_dense_cfg.json", "k_dense_walk|k_dense_gg", "--dense --frames 300 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing"
    correction = ("MI355X_MICROARCH.md HBM section: the doubling of FETCH_SIZE applies to wide coalesced 16 B/lane streams (128 B requests "
                  "tallied at 64 B); this kernel requests 64 B lines (8 lanes x 8 B of flow per run), 32 B of source depth, 8 mask bytes and "
                  "scattered 4 B target depths: taken as is, as is WRITE_SIZE (both uncalibrated for these widths); Infinity-Cache hits are counted")
    files = [f"profiles/{tag}_pmc_FETCH_SIZE_dense.csv", f"profiles/{tag}_pmc_WRITE_SIZE_dense.csv"]
else:
    kernel, factor = "k_matvec_pairs_fast<4", 2.0  # (any workgroup-size instantiation)
    f, nf = avg(os.path.join(root, "pmc_fetch"), "FETCH_SIZE", kernel)
    w, nw = avg(os.path.join(root, "pmc_write"), "WRITE_SIZE", kernel)
    bench_file, cmd_kernel, cmd_bench = "bench.json", "k_matvec_pairs_fast", "--steps 4 --warmup 1 --no-cpu-baseline --no-secondary --pcg-lockstep (lockstep: no early-exit launches in the average)"
    correction = ("MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE counts a wide coalesced 16 B/lane stream at "
                  "half its bytes -> doubled; WRITE_SIZE uncalibrated, taken as is")
    files = [f"profiles/{tag}_pmc_FETCH_SIZE.csv", f"profiles/{tag}_pmc_WRITE_SIZE.csv"]
bench_line = json.loads([l for l in open(os.path.join(root, bench_file)) if l.startswith("{")][-1])
out = {
    "kernel": kernel,
    "kernel_sources_sha256": bench.kernel_sources_digest(dense=dense),
    "constraints": bench_line["config"]["constraints"], "pairs": bench_line["config"]["pairs"],
    "command": f"rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --kernel-include-regex \"{cmd_kernel}\" "
               f"--output-format csv -- python bench.py {cmd_bench}",
    "fetch_size_kb_per_launch_raw": f, "write_size_kb_per_launch_raw": w, "dispatches": nf,
    "correction": correction,
    "traffic_bytes_per_launch": (factor * f + w) * 1024.0,
    "source_files": files,
}
print(json.dumps(out, indent=1))
