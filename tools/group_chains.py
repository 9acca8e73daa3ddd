#!/usr/bin/env python3
"""The env groups' kernel chains, from a rocprofv3 kernel trace of bench.py in its default mode (tools/collect_profiles.sh leaves one under gpurun_out/prof_<tag>/trace_default):
per group (= launch size) the average duration of each kernel of its chain over the last 50 grouped steps, the chain's sum, and the step period.  Says whether a step is the
chain of ONE group (latency of its heaviest env) or the machine shared between balanced chains (throughput).

    python tools/group_chains.py gpurun_out/prof_r05/trace_default > profiles/r05_group_chains.txt"""
import collections
import csv
import glob
import os
import sys

import numpy as np

root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_r05/trace_default'
f = sorted(glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
STEP = ('k_prep2', 'k_solve2', 'k_action_prep', 'k_calc_state', 'k_member')
ks = [r for r in rows if r['Kernel_Name'].split('(')[0] in STEP]
mem = [i for i, r in enumerate(ks) if r['Kernel_Name'].startswith('k_member')]
blocks = lambda r: int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])   # noqa: E731
n_envs = max(blocks(r) for r in ks if r['Kernel_Name'].startswith('k_prep2'))
sel = [(a, b) for a, b in zip(mem[:-1], mem[1:]) if any(r['Kernel_Name'].startswith('k_prep2') and blocks(r) < n_envs for r in ks[a:b])][-50:]
by = collections.defaultdict(list)
queue_of = {}
for a, b in sel:
    for r in ks[a:b]:
        name = r['Kernel_Name'].split('(')[0]
        by[(name, blocks(r))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        queue_of[(name, blocks(r))] = r['Queue_Id']
period = np.median(np.diff([int(ks[a]['Start_Timestamp']) for a, b in sel])) / 1e3
print('%s\n%d grouped steps, %d envs; step period (k_member to k_member, median) %.1f us' % (f, len(sel), n_envs, period))
preps = sorted(g for (n, g) in by if n == 'k_prep2')
acts = sorted(g for (n, g) in by if n == 'k_action_prep')
solves = sorted(g for (n, g) in by if n == 'k_solve2')
calcs = sorted(g for (n, g) in by if n == 'k_calc_state')
print('group | envs | queue | k_action_prep | k_prep2 x 11: mean p50 p90 max | k_solve2 x 12: mean p50 p90 max | k_calc_state | chain sum [us]')
for gi, g in enumerate(preps):
    so, ca, pr = np.array(by[('k_solve2', solves[gi])]), np.array(by[('k_calc_state', calcs[gi])]), np.array(by[('k_prep2', g)])
    if len(acts) == len(preps):      # every group launches its own k_action_prep
        ap = np.array(by[('k_action_prep', acts[gi])])
        chain, first = ap.mean() + 11 * pr.mean() + 12 * so.mean() + ca.mean(), '%.1f' % ap.mean()
    else:                            # (an experiment of round 5) the IK of all envs in group 0's launch; the other groups start with a plain k_prep2 and wait for it before their first k_solve2
        ap = np.array(by[('k_action_prep', acts[0])])
        chain = max(ap.mean(), pr.mean()) + 11 * pr.mean() + 12 * so.mean() + ca.mean()
        first = '%.1f%s' % (ap.mean(), '' if gi == 0 else ' (waited for)')
    print('%d | %d | %s | %s | %.1f %.1f %.1f %.1f | %.1f %.1f %.1f %.1f | %.1f | %.0f' % (gi, g, queue_of[('k_prep2', g)], first, pr.mean(), np.median(pr), np.percentile(pr, 90), pr.max(),
                                                                                    so.mean(), np.median(so), np.percentile(so, 90), so.max(), ca.mean(), chain))
one = [(a, b) for a, b in zip(mem[:-1], mem[1:]) if all(blocks(r) >= n_envs or not r['Kernel_Name'].startswith('k_prep2') for r in ks[a:b]) and any(r['Kernel_Name'].startswith('k_prep2') for r in ks[a:b])][-30:]
if one:
    d = collections.defaultdict(list)
    for a, b in one:
        for r in ks[a:b]:
            d[r['Kernel_Name'].split('(')[0]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    print('the same kernels with all envs in one group (%d steps of the timers region): %s' % (len(one), {k: round(float(np.mean(v)), 1) for k, v in d.items()}))
