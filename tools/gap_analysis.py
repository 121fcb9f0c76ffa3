"""Development aid: GPU idle gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV."""
import collections, csv, glob, sys
fn = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(fn)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('cvd::', '')) for r in rows)
print('span ms %.3f  busy ms %.3f' % ((ev[-1][1] - ev[0][0]) / 1e6, sum(e - s for s, e, _ in ev) / 1e6))
gaps = collections.Counter(); gapn = collections.Counter()
for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
    g = s1 - e0
    if g > 0:
        gaps[(n0[:28], n1[:28])] += g; gapn[(n0[:28], n1[:28])] += 1
print('total gap ms %.3f' % (sum(gaps.values()) / 1e6))
for k, v in gaps.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 16):
    print('%-30s -> %-30s  n=%4d  total %.3f ms  avg %.1f us' % (k[0], k[1], gapn[k], v / 1e6, v / gapn[k] / 1e3))
