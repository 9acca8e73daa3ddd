#!/bin/bash
# Round 4: the bench at other batch sizes and with the contact-model switches, for DESIGN.md section 6.   gpurun -- 'bash tools/r04_sizes.sh'
OUT=gpurun_out/r04_sizes
mkdir -p $OUT
line() { python - "$1" <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d = json.loads(l)
        print('%s: %.3f M first region, median %.3f M, %.3f ms/step, k_prep2 %.4f k_solve2 %.4f k_action %.4f ms' % (sys.argv[1].split('/')[-1], d['value'] / 1e6, d['repeats']['median'] / 1e6, d['ms_per_step'],
              d['roofline']['per_launch_ms']['k_prep2'], d['roofline']['per_launch_ms']['k_solve2'], d['roofline']['per_launch_ms']['k_action']))
PY
}
for n in 1024 2048 8192 16384 32768; do python bench.py --envs-per-gpu $n --no-cpu-baseline --no-extras > $OUT/n$n.json 2>/dev/null; line $OUT/n$n.json; done
RP_NO_GJK=1 python bench.py --no-cpu-baseline > $OUT/obb_edges.json 2>/dev/null; line $OUT/obb_edges.json
python bench.py --no-cpu-baseline --no-extras --stateless-contacts > $OUT/stateless.json 2>/dev/null; line $OUT/stateless.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/driver_flags.json 2>/dev/null; line $OUT/driver_flags.json
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r04_sizes/obb_edges.json') if l.startswith('{')][0])
print('RP_NO_GJK distribution A: %.3f M' % (d['distribution_A']['value'] / 1e6))
PY
