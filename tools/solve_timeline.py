#!/usr/bin/env python3
"""Timeline of ONE timed solve of bench.py from a rocprofv3 --kernel-trace CSV: every kernel of the main queue between two host-to-device
uploads of the start state (the timed solves restart from the level's start state), in start order with durations and the idle gaps
before them, the PCG loop's launches folded.  Shows what a solve pays besides its LM iterations (set-up, first evaluation, level builds).
usage: solve_timeline.py <trace dir> [which solve from the end, default 2]"""
import csv, glob, sys

root = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
fn = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(fn)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cvd::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
# a timed solve starts with an assembly (the first evaluation) that follows host-to-device uploads of the start state and no candidate-cost pass
asm = [i for i, r in enumerate(rows) if r[2].startswith(("k_assemble_fast", "k_dense_walk")) and (r[1] - r[0]) / 1e3 > 250]
starts = []
for i in asm:
    back = [rows[j][2] for j in range(max(0, i - 12), i)]
    if sum(1 for b in back if "copyBuffer" in b) >= 2 and not any(b.startswith("k_cost_items") for b in back):
        starts.append(max(0, i - 14))
if len(starts) < which + 1:
    print("not enough solves in the trace:", len(starts))
    sys.exit(0)
a, b = starts[-which - 1], starts[-which]
seg = rows[a:b]
t0 = seg[0][0]
print(f"# solve of {len(seg)} launches, span {(seg[-1][1] - t0) / 1e3:.1f} us (+ {(rows[b][0] - seg[-1][1]) / 1e3:.1f} us idle before the next solve's first kernel)")
i = 0
while i < len(seg):
    s, e, n = seg[i]
    if n.startswith("k_matvec_pairs") or n.startswith("k_pcg_tail") or n.startswith("k_cross_matvec"):
        j = i
        while j < len(seg) and (seg[j][2].startswith("k_matvec_pairs") or seg[j][2].startswith("k_pcg_tail") or seg[j][2].startswith("k_cross_matvec")
                                or seg[j][2].startswith("k_matvec_finish") or seg[j][2].startswith("k_cg_update") or seg[j][2].startswith("k_coarse_apply")):
            j += 1
        busy = sum(x[1] - x[0] for x in seg[i:j])
        print(f"{(s - t0) / 1e3:10.1f} {(seg[j - 1][1] - s) / 1e3:9.1f}           PCG loop: {j - i} launches, busy {busy / 1e3:.1f} us")
        i = j
        continue
    gap = (s - seg[i - 1][1]) / 1e3 if i else 0.0
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {gap:9.1f}  {n[:60]}")
    i += 1
