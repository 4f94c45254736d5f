#!/bin/bash
# HBM traffic of one heuristic relaxation (16384^2): separate FETCH_SIZE / WRITE_SIZE passes, summed
# over every kernel of one bench step (GPU box).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/wtraffic; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $O/$c -- python3 $R/bench.py --workload wavefront --cells 16384 --steps 1 --warmup 0 --no-cpu-baseline > $O/$c.log 2>&1 || { echo "$c failed"; tail -3 $O/$c.log; exit 1; }
done
python3 - <<'PY'
import csv,glob,os,collections
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/wtraffic"
tot={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=sorted(glob.glob(f"{O}/{c}/*/*counter_collection.csv"), key=os.path.getmtime)[-1]
    by=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]==c: by[r["Kernel_Name"].split("(")[0][-24:]]+=float(r["Counter_Value"])
    tot[c]=sum(by.values())
    print(c, {k: round(v/1e6,2) for k,v in by.items()}, "(M KiB units)")
print("HBM bytes per relaxation = (2*FETCH + WRITE)*1024 =", (2*tot["FETCH_SIZE"]+tot["WRITE_SIZE"])*1024)
PY
