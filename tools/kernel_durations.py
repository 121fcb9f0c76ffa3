#!/usr/bin/env python3
"""profiles/<tag>_kernel_durations.txt from a rocprofv3 --kernel-trace CSV: per kernel launches / total / mean / median and
the mean over "working launches" (durations above half the median: the PCG kernels enqueued past convergence return at
once and would otherwise pull the --stats average down).  usage: kernel_durations.py <trace dir> <tag>"""
import collections, csv, glob, statistics, sys

root, tag = sys.argv[1], sys.argv[2]
fn = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(fn)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cvd::", "")
    d[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"# per-kernel durations from rocprofv3 --kernel-trace of `python bench.py --steps 20 --warmup 3 --no-cpu-baseline`")
print(f"# (the --stats averages in {tag}_bench_kernel_stats.csv include the early-exit PCG launches enqueued past convergence;")
print(f"#  \"working launches\" excludes them: durations above half the median)")
for name, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:30]:
    med = statistics.median(v)
    w = [x for x in v if x > med / 2]
    print(f"{name:34s} launches {len(v):5d}  total {sum(v) / 1e3:8.3f} ms  mean {sum(v) / len(v):8.1f} us  median {med:8.1f} us | "
          f"working launches (> median/2) {len(w):5d}  mean {sum(w) / len(w):8.1f} us")
