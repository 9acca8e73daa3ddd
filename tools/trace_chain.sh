#!/bin/bash
# GPU box: kernel trace of a short bench run, then the timeline of one step (per stream: kernel, start offset, duration, gap to the previous kernel)
#   bash tools/trace_chain.sh [groups] [extra bench args]
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
G=${1:-3}
OUT=$REPO/gpurun_out/trace_chain_g$G
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/t" -- python3 "$REPO/bench.py" --steps 12 --warmup 4 --no-cpu-baseline --no-extras --repeats 1 --groups $G > "$OUT/bench.json" 2> "$OUT/log.txt"
cd "$REPO"
python3 tools/trace_chain.py "$OUT/t" > "$OUT/chain.txt"
grep -E "^step|^---|total gaps" "$OUT/chain.txt"
