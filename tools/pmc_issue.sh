#!/bin/bash
# Runs on the GPU box: instruction-issue counters of the step kernels (one group, so kernels do not overlap).
#   gpurun --timeout 900 -- 'bash tools/pmc_issue.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_issue
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters.txt" 2>&1
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/$tag" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --groups 1 --no-cpu-baseline --no-extras --repeats 1 > /dev/null 2> "$OUT/$tag.log"
done
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        a = acc[k][r['Counter_Name']]
        a[0] += float(r['Counter_Value']); a[1] += 1
with open(out + '/summary.txt', 'w') as o:
    for k, cs in sorted(acc.items()):
        if not k.startswith('k_'): continue
        o.write(k + '\n')
        for c, (s, n) in sorted(cs.items()):
            o.write('  %-28s per launch %14.1f  (launches %d)\n' % (c, s / n, n))
print(open(out + '/summary.txt').read())
PY
