#!/usr/bin/env python3
"""Timeline view of a rocprofv3 --kernel-trace CSV: for the main stream's kernels in start order, the idle gap before each
kernel, aggregated per (previous kernel -> kernel) transition, and the busy / idle split of the traced interval.
usage: kernel_gaps.py <trace dir> [first_fraction last_fraction]"""
import collections, csv, glob, sys

root = sys.argv[1]
lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.5, 1.0)
fn = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(fn)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cvd::", "").split("<")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "0")))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
a, b = t0 + lo * (t1 - t0), t0 + hi * (t1 - t0)
rows = [r for r in rows if a <= r[0] <= b]
queues = collections.Counter(r[3] for r in rows)
main = queues.most_common(1)[0][0]
print("queues:", dict(queues), "main:", main)
rows = [r for r in rows if r[3] == main]
busy = sum(e - s for s, e, _, _ in rows)
span = rows[-1][1] - rows[0][0]
print(f"span {span / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms ({100.0 * busy / span:.1f} %)  kernels {len(rows)}")
gaps = collections.defaultdict(list)
for p, c in zip(rows, rows[1:]):
    gaps[(p[2], c[2])].append((c[0] - p[1]) / 1e3)
print("transition                                                            count   mean gap us   total ms")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:24]:
    print(f"{k[0][:32]:32s} -> {k[1][:32]:32s} {len(v):6d} {sum(v) / len(v):10.2f} {sum(v) / 1e3:10.3f}")
# every idle gap above 30 us of the last fifth of the window, with the kernels around it
tail = rows[int(len(rows) * 0.8):]
print("gaps > 30 us (time since the window's start in ms, gap in us, previous kernel -> next kernel [duration us])")
for i in range(1, len(tail)):
    gap = (tail[i][0] - tail[i - 1][1]) / 1e3
    if gap > 30.0:
        print(f"  t {((tail[i][0] - tail[0][0]) / 1e6):9.3f}  gap {gap:8.1f}   {tail[i - 1][2][:28]:28s} [{(tail[i - 1][1] - tail[i - 1][0]) / 1e3:7.1f}] -> {tail[i][2][:28]:28s} [{(tail[i][1] - tail[i][0]) / 1e3:7.1f}]")
