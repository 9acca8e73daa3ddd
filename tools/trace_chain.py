#!/usr/bin/env python3
"""timeline of one rp_step from a rocprofv3 --kernel-trace CSV: per queue/stream the kernels with start offset, duration and the gap
to the previous kernel of the same queue; plus totals.  Usage: trace_chain.py <dir>"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True), key=os.path.getmtime)[-1]      # the newest run
rows = list(csv.DictReader(open(f)))
ks = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0], r.get('Queue_Id', r.get('Stream_Id', '0'))) for r in rows]
ks.sort()
# find the last k_member launches: each marks the start of a step
starts = [i for i, k in enumerate(ks) if k[2].startswith('k_member')]
if len(starts) < 3:
    print('no steps found'); sys.exit(0)
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = starts[-k], starts[-k + 1]
step = ks[a:b]
t0 = step[0][0]
print('step: %d kernels, %.3f ms from first start to last end' % (len(step), (max(k[1] for k in step) - t0) / 1e6))
byq = {}
for k in step:
    byq.setdefault(k[3], []).append(k)
for q, lst in byq.items():
    print('--- queue', q, len(lst), 'kernels; busy %.3f ms; span %.3f ms' % (sum(k[1] - k[0] for k in lst) / 1e6, (lst[-1][1] - lst[0][0]) / 1e6))
    prev = None
    gaps = 0
    for k in lst:
        gap = (k[0] - prev) / 1e3 if prev else 0.0
        gaps += max(gap, 0)
        print('  %-16s start %8.1f us  dur %7.1f us  gap %6.1f us' % (k[2][:16], (k[0] - t0) / 1e3, (k[1] - k[0]) / 1e3, gap))
        prev = k[1]
    print('  total gaps %.1f us' % gaps)
