#!/usr/bin/env python3
"""Where an LM iteration's wall time goes, from a rocprofv3 --kernel-trace CSV of bench.py: the trace is cut at the k_lm_diag
launches (one per LM iteration); for the LM iterations of the last `--last` fraction of the trace it prints (a) the mean split
PCG loop / before it / after it, busy and idle, (b) one iteration (the median-length one) kernel by kernel with the PCG loop folded.
usage: lm_timeline.py <trace dir> [--last 0.4] [--full]"""
import csv, glob, statistics, sys

root = sys.argv[1]
last = float(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0.4
full = "--full" in sys.argv
fn = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(fn)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cvd::", "")
    name = name.split("<")[0] if name.startswith("rocprim") or name.startswith("__amd") else name
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
rows = [r for r in rows if r[0] >= t1 - last * (t1 - t0)]
cuts = [i for i, r in enumerate(rows) if r[2].startswith("k_lm_diag")]
its = [rows[a:b] for a, b in zip(cuts, cuts[1:])]
its = [it for it in its if sum(1 for r in it if r[2].startswith("k_matvec_pairs")) >= 5]
if not its:
    sys.exit("no LM iterations found")
PCG = ("k_matvec_pairs", "k_pcg_tail", "k_cg_update", "k_matvec_finish", "k_cross_matvec", "k_coarse_apply", "k_coarse_dense_apply",
       "k_tl_rows", "k_pcg_scalars")


def split(it, nxt_start):
    first = next(i for i, r in enumerate(it) if r[2].startswith("k_matvec_pairs") or r[2].startswith("k_cross_matvec"))
    lastp = max(i for i, r in enumerate(it) if r[2].startswith(PCG))
    seg = {"pre": it[:first], "pcg": it[first:lastp + 1], "post": it[lastp + 1:]}
    bounds = {"pre": (it[0][0], it[first][0]), "pcg": (it[first][0], it[lastp][1]), "post": (it[lastp][1], nxt_start)}
    return seg, bounds


acc = {k: [0.0, 0.0] for k in ("pre", "pcg", "post")}
npcg, spans = [], []
for k, it in enumerate(its):
    nxt = its[k + 1][0][0] if k + 1 < len(its) else it[-1][1]
    seg, b = split(it, nxt)
    for key in acc:
        acc[key][0] += (b[key][1] - b[key][0]) / 1e3
        acc[key][1] += sum(e - s for s, e, _ in seg[key]) / 1e3
    npcg.append(sum(1 for r in seg["pcg"] if r[2].startswith("k_matvec_pairs") or r[2].startswith("k_cross_matvec")))
    spans.append((nxt - it[0][0]) / 1e3)
n = len(its)
print(f"# {n} LM iterations, mean span {sum(spans) / n:.1f} us, mean PCG products per iteration {sum(npcg) / n:.1f}")
for key, label in (("pre", "before the PCG (k_lm_diag .. first product)"), ("pcg", "PCG loop"), ("post", "after the PCG (.. next k_lm_diag)")):
    print(f"{label:48s} span {acc[key][0] / n:8.1f} us   kernels busy {acc[key][1] / n:8.1f} us   idle {(acc[key][0] - acc[key][1]) / n:8.1f} us")
mid = sorted(range(n), key=lambda i: spans[i])[n // 2]
it = its[mid]
nxt = its[mid + 1][0][0] if mid + 1 < n else it[-1][1]
seg, b = split(it, nxt)
print(f"# iteration {mid} (median span {spans[mid]:.1f} us): start offset us, duration us, idle before us, kernel")
prev_end = it[0][0]


def show(rs):
    global prev_end
    for s, e, name in rs:
        print(f"  {(s - it[0][0]) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}  {name}")
        prev_end = max(prev_end, e)


show(seg["pre"])
if full:
    show(seg["pcg"])
else:
    by = {}
    for s, e, name in seg["pcg"]:
        by.setdefault(name, []).append((e - s) / 1e3)
    span = (b["pcg"][1] - b["pcg"][0]) / 1e3
    busy = sum(sum(v) for v in by.values())
    print(f"  {(b['pcg'][0] - it[0][0]) / 1e3:9.1f} {span:8.1f}          PCG loop: {npcg[mid]} products, busy {busy:.1f} us, idle {span - busy:.1f} us")
    for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print(f"                                   {len(v):3d} x {statistics.median(v):7.1f} us (median)  {name}")
    prev_end = b["pcg"][1]
show(seg["post"])
print(f"  {(nxt - it[0][0]) / 1e3:9.1f}                   next k_lm_diag")
