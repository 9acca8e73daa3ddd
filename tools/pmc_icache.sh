#!/bin/bash
# GPU box: instruction-cache counters of the step kernels, one group (kernels alone) and three groups (kernels of different groups side by side)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_icache
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for g in 1 3; do
  timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d "$OUT/g$g" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --groups $g --no-cpu-baseline --no-extras --repeats 1 > /dev/null 2> "$OUT/g$g.log"
  timeout 300 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d "$OUT/h$g" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --groups $g --no-cpu-baseline --no-extras --repeats 1 > /dev/null 2> "$OUT/h$g.log"
done
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
for g in ('g1', 'g3', 'h1', 'h3'):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(out + '/' + g + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            a = acc[k][r['Counter_Name']]
            a[0] += float(r['Counter_Value']); a[1] += 1
    print('==', g)
    for k in ('k_prep2', 'k_solve2', 'k_action_prep', 'k_calc_state'):
        if k in acc:
            print(' ', k, {c: round(s / n) for c, (s, n) in sorted(acc[k].items())})
PY
