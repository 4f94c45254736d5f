#!/bin/bash
# Kernel-trace profile of the heuristic relaxation at one size / band (GPU box).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/wprof; mkdir -p "$O"
N=${CELLS:-16384}; B=${BAND:-128}
export TMPDIR=/tmp
cd /tmp || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/p" -- python3 "$R/bench.py" --workload wavefront --cells "$N" --steps 2 --warmup 1 --no-cpu-baseline --band "$B" > "$O/run.log" 2>&1 || { tail -20 "$O/run.log"; exit 1; }
f=$(find "$O/p" -name "*kernel_stats.csv" | head -1)
cat "$f"
