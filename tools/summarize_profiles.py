#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/collect_profiles.sh into the small files kept under profiles/:
   <tag>_kernel_stats_g1.csv / <tag>_kernel_stats_default.csv (per-kernel call counts and durations) and
   <tag>_pmc_summary.json (per-kernel FETCH_SIZE / WRITE_SIZE averages per launch, KB as rocprofv3 reports them)."""
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(src, 'summary')
os.makedirs(dst, exist_ok=True)


def find(sub, pattern):
    hits = sorted(glob.glob(os.path.join(src, sub, '**', pattern), recursive=True))
    return hits[0] if hits else None


for sub, name in (('trace_g1', 'kernel_stats_g1'), ('trace_default', 'kernel_stats_default')):
    f = find(sub, '*kernel_stats.csv')
    if f:
        shutil.copy(f, os.path.join(dst, '%s_%s.csv' % (tag, name)))
    else:
        print('missing kernel stats for', sub)

pmc = {}
for sub, counter in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
    f = find(sub, '*counter_collection.csv')
    if not f:
        print('missing counter collection for', sub)
        continue
    acc = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row.get('Counter_Name') != counter:
                continue
            k = row['Kernel_Name'].split('(')[0]
            acc.setdefault(k, {})
            acc[k].setdefault(row['Dispatch_Id'], 0.0)
            acc[k][row['Dispatch_Id']] += float(row['Counter_Value'])
    for k, d in acc.items():
        vals = list(d.values())
        e = pmc.setdefault(k, {})
        e['launches_sampled'] = len(vals)
        e['%s_KB_avg' % counter] = sum(vals) / len(vals)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from roboticsplayroompybullet_amd import _lib  # noqa: E402
pmc['library_version'] = _lib.load().rp_version().decode()      # bench.py refuses to quote a profile of another library version
pmc['command'] = 'bench.py --steps 5 --warmup 1 --groups 1 --no-cpu-baseline --no-extras (one rocprofv3 --pmc pass per counter)'
json.dump(pmc, open(os.path.join(dst, '%s_pmc_summary.json' % tag), 'w'), indent=1)
print(json.dumps({k: v for k, v in pmc.items() if k in ('k_solve2', 'k_prep2')}, indent=1))
