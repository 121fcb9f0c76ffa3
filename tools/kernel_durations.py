#!/usr/bin/env python3
"""profiles/<tag>_kernel_durations.txt from a rocprofv3 --kernel-trace CSV: per kernel launches / total / mean / median and
the mean over "working launches" (durations above half the median: the PCG kernels enqueued past convergence return at
once and would otherwise pull the --stats average down).  Launches are bucketed by GRID SIZE as well (a pipeline's
coarse-to-fine levels launch the same kernel with different grids: the final level's rows are the largest grids).
usage: kernel_durations.py <trace dir> <tag> "<the profiled command line>" """
import collections, csv, glob, statistics, sys

root, tag = sys.argv[1], sys.argv[2]
cmdline = sys.argv[3] if len(sys.argv) > 3 else "(command line not recorded)"
fn = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
bygrid = collections.defaultdict(list)
when = collections.defaultdict(list)
for r in csv.DictReader(open(fn)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cvd::", "")
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    d[name].append(dur)
    when[name].append((int(r["Start_Timestamp"]), dur))
    grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
    lds = r.get("LDS_Block_Size") or r.get("LDS_Block_Size_v") or ""
    if lds and int(lds) > 65536:   # (the dense mode's walks launch one grid at every coarse-to-fine level: their LDS size tells the levels apart)
        grid = f"{grid}/lds{lds}"
    bygrid[(name, grid)].append(dur)
print(f"# per-kernel durations from rocprofv3 --kernel-trace of `{cmdline}`")
print(f"# (the --stats averages in {tag}_bench_kernel_stats.csv include the early-exit PCG launches enqueued past convergence;")
print(f"#  \"working launches\" excludes them: durations above half the median)")
for name, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:30]:
    med = statistics.median(v)
    w = [x for x in v if x > med / 2]
    print(f"{name:34s} launches {len(v):5d}  total {sum(v) / 1e3:8.3f} ms  mean {sum(v) / len(v):8.1f} us  median {med:8.1f} us | "
          f"working launches (> median/2) {len(w):5d}  mean {sum(w) / len(w):8.1f} us")
print("# the same, per grid size (threads), for the five kernels with the largest totals")
for name, _ in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:5]:
    for (n2, grid), v in sorted(((k, v) for k, v in bygrid.items() if k[0] == name), key=lambda kv: -sum(kv[1]))[:4]:
        med = statistics.median(v)
        w = [x for x in v if x > med / 2]
        print(f"{name:34s} grid {grid:>18s} launches {len(v):5d}  total {sum(v) / 1e3:8.3f} ms  median {med:8.1f} us  working mean {sum(w) / len(w):8.1f} us")
print("# ... and their LAST six working launches in time order (the end of the trace is the timed region: the final coarse-to-fine level,")
print("#     which is what bench.py's HIP events time; a dense-mode kernel runs at every level of the pipeline with the same grid)")
for name, _ in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:5]:
    med = statistics.median(d[name])
    last = [x for _, x in sorted(when[name]) if x > med / 2][-6:]
    print(f"{name:34s} " + "  ".join(f"{x:9.1f}" for x in last) + f"  us   mean {sum(last) / len(last):9.1f} us")
