#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace statistics and the two PMC passes of bench.py, then the
# summaries that get committed under profiles/.  Counters are collected in their own runs (never with --sys-trace /
# --runtime-trace), and the profiled program sits directly after `--`.
#   gpurun --timeout 900 -- 'bash tools/collect_profiles.sh r01'
set -u
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_g1" -- python3 "$REPO/bench.py" --steps 50 --warmup 5 --groups 1 --no-cpu-baseline --no-extras --repeats 1 > "$OUT/bench_g1.json" 2> "$OUT/trace_g1.log"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_default" -- python3 "$REPO/bench.py" --steps 50 --warmup 5 --no-cpu-baseline --no-extras --repeats 1 > "$OUT/bench_default_profiled.json" 2> "$OUT/trace_default.log"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --groups 1 --no-cpu-baseline --no-extras --repeats 1 > /dev/null 2> "$OUT/pmc_fetch.log"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --groups 1 --no-cpu-baseline --no-extras --repeats 1 > /dev/null 2> "$OUT/pmc_write.log"
cd "$REPO"
python3 tools/summarize_profiles.py "$OUT" "$TAG"
cp "$OUT/summary/${TAG}_pmc_summary.json" profiles/ 2>/dev/null      # (on the box: so that the bench lines below quote this build's traffic; the copy in gpurun_out/ is what travels back)
timeout 300 python3 bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_line.log"
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_flags.json" 2>> "$OUT/bench_line.log"
tail -c 600 "$OUT/bench_line.json"
