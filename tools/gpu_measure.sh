#!/bin/bash
# Measurement pass on the GPU box: smoke, the three bench workloads, a kernel-trace profile of the
# default bench command and the PMC passes for HBM traffic.  Every GPU step has its own time
# limit and the script stops at the first failure.  Outputs land in gpurun_out/measure/.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/measure
mkdir -p "$O"
cd "$R" || exit 1
step() {  # name, seconds, command...
  local name=$1 secs=$2
  shift 2
  echo "== $name"
  timeout -k 10 "$secs" "$@" >"$O/$name.log" 2>&1
  local rc=$?
  echo "   rc=$rc"
  if [ $rc -ne 0 ]; then
    tail -20 "$O/$name.log"
    exit $rc
  fi
}
step smoke 300 python3 -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
step bench_collision 300 python3 -u bench.py
tail -1 "$O/bench_collision.log"
step bench_rs 300 python3 -u bench.py --workload rs --steps 10
tail -1 "$O/bench_rs.log"
step bench_wave4096 300 python3 -u bench.py --workload wavefront --cells 4096 --steps 3 --warmup 1
tail -1 "$O/bench_wave4096.log"
step bench_wave16384 600 python3 -u bench.py --workload wavefront --cells 16384 --steps 3 --warmup 1 --no-cpu-baseline
tail -1 "$O/bench_wave16384.log"
step bench_expand 300 python3 -u bench.py --workload expand --steps 10
tail -1 "$O/bench_expand.log"
step bench_analytic 300 python3 -u bench.py --workload analytic --steps 5 --warmup 1
tail -1 "$O/bench_analytic.log"
step bench_queries 600 python3 -u bench.py --workload queries --steps 3 --warmup 1
tail -1 "$O/bench_queries.log"
export TMPDIR=/tmp
cd /tmp || exit 1
step prof_collision 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_collision" -- python3 "$R/bench.py" --no-cpu-baseline
step prof_rs 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_rs" -- python3 "$R/bench.py" --workload rs --steps 10 --no-cpu-baseline
step prof_wave 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_wave" -- python3 "$R/bench.py" --workload wavefront --cells 16384 --steps 2 --warmup 1 --no-cpu-baseline
step prof_queries 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_queries" -- python3 "$R/bench.py" --workload queries --steps 2 --warmup 1 --no-cpu-baseline
# HBM traffic: separate counter passes, no tracing domains
step pmc_fetchsize 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 "$R/bench.py" --no-cpu-baseline --steps 3 --warmup 1
step pmc_writesize 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 "$R/bench.py" --no-cpu-baseline --steps 3 --warmup 1
find "$O" -name "*.csv" | head -40
echo done
