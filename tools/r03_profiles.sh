#!/bin/bash
# Evidence for the profiles directory (GPU box).  One bench line and one `rocprofv3 --kernel-trace --stats` run per
# workload, each with --no-extra so that a CSV holds only that leg's launches, plus the counter
# passes: the planner's fused round kernel (k_round) on the search-heavy leg, the Reeds-Shepp batch
# kernel's instruction-cache and issue counters, and the collision kernel's HBM bytes.
# Outputs: gpurun_out/r03prof/<leg>.json, <leg>_kernel_stats.csv, pmc_*/summary.txt.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03prof; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
declare -A ARGS=(
  [collision]="--workload collision --no-extra"
  [rs]="--workload rs --steps 10 --warmup 2"
  [wavefront]="--workload wavefront --steps 2 --warmup 1"
  [queries]="--workload queries --steps 6 --warmup 2"
  [queries_heavy]="--workload queries --heavy --steps 2 --warmup 1"
  [expand]="--workload expand --steps 10 --warmup 2"
  [analytic]="--workload analytic --steps 5 --warmup 1"
)
for leg in ${LEGS:-collision rs wavefront queries queries_heavy expand analytic}; do
  a=${ARGS[$leg]}
  cd $R && timeout -k 10 400 python3 bench.py $a > $O/$leg.log 2>&1 || { echo "$leg bench failed"; tail -5 $O/$leg.log; exit 1; }
  tail -n 1 $O/$leg.log > $O/$leg.json
  cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$leg -- python3 $R/bench.py $a --no-cpu-baseline --no-floors > $O/trace_$leg.log 2>&1 || { echo "$leg trace failed"; tail -5 $O/trace_$leg.log; exit 1; }
  f=$(find $O/trace_$leg -name "*kernel_stats.csv" | head -1); cp $f $O/${leg}_kernel_stats.csv
  rm -rf $O/trace_$leg
  python3 - $O/$leg.json $O/${leg}_kernel_stats.csv <<'PY'
import csv, json, sys
d = json.loads(open(sys.argv[1]).read())
print(f'{d["metric"]:38s} {d["value"]:.4g} {d["unit"]}  {d["ms_per_step"]:.3f} ms/step  roofline {d["roofline"].get("bound")} frac {d["roofline"].get("frac")}')
for r in list(csv.DictReader(open(sys.argv[2])))[:4]:
    print("   ", r["Name"][:70].ljust(70), "calls", r["Calls"], "avg_ns", r["AverageNs"])
PY
done
[ -n "${NO_PMC:-}" ] && exit 0
OCC="occ:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
INS="ins:SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_IFETCH"
bash $R/tools/pmc_kernels.sh r03prof/pmc_queries_heavy "--workload queries --heavy --steps 1 --warmup 1" "$OCC" "$INS" > /dev/null || exit 1
bash $R/tools/pmc_kernels.sh r03prof/pmc_rs "--workload rs --steps 1 --warmup 1" "$OCC" "$INS" > /dev/null || exit 1
bash $R/tools/pmc_kernels.sh r03prof/pmc_collision "--workload collision --steps 2 --warmup 1" "fetch:FETCH_SIZE" "write:WRITE_SIZE" > /dev/null || exit 1
# the raw per-launch CSVs are tens of megabytes; the summaries are what is kept
for t in queries_heavy rs collision; do find $O/pmc_$t -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +; done
for t in queries_heavy rs collision; do echo "== pmc_$t"; grep -E "k_round|k_rs_batch|k_collision|k_expand |k_ftab" $O/pmc_$t/summary.txt | head -40; done
