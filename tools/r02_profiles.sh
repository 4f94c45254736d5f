#!/bin/bash
# The default bench line (every config under "extra") and the kernel-trace statistics of the same command (GPU box).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02prof; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
cd $R && timeout -k 10 600 python3 bench.py > $O/bench_default.log 2>&1 || { tail -5 $O/bench_default.log; exit 1; }
tail -n 1 $O/bench_default.log > $O/bench_default.json
cd $R && timeout -k 10 300 python3 bench.py --workload queries --heavy --steps 3 --warmup 1 > $O/bench_queries_heavy.log 2>&1 || { tail -5 $O/bench_queries_heavy.log; exit 1; }
tail -n 1 $O/bench_queries_heavy.log > $O/bench_queries_heavy.json
cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; head -20 $O/kernel_stats.csv
