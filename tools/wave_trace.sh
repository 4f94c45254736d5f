#!/bin/bash
# Kernel timeline of one heuristic relaxation (GPU box): per-launch start / end of the list and round kernels.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/wtrace; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/p -- python3 $R/bench.py --workload wavefront --cells ${CELLS:-16384} --steps 1 --warmup 0 --no-cpu-baseline --map-kind ${KIND:-bernoulli} > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
f=$(find $O/p -name "*kernel_trace.csv" | head -1)
python3 - $f > $O/timeline.txt <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "k_tile" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"]); prev=None
for r in rows:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    name=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0].split("<")[0]
    print(f"{name:14s} start {(s-t0)/1e3:10.1f} us dur {(e-s)/1e3:8.1f} gap {((s-prev)/1e3 if prev else 0):6.1f}")
    prev=e
PY
python3 - $O/timeline.txt <<'PY'
import sys
L=[l.split() for l in open(sys.argv[1])]
for k in ("k_tile_list","k_tile_round"):
    d=[float(l[4]) for l in L if l[0]==k]; g=[float(l[6]) for l in L if l[0]==k]
    print(k,"n",len(d),"dur sum %.1f us mean %.1f"%(sum(d),sum(d)/len(d)),"gap-before sum %.1f mean %.2f"%(sum(g),sum(g)/len(g)))
PY
