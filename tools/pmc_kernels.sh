#!/bin/bash
# Counter passes over one bench workload, summed per kernel (GPU box).
# usage: tools/pmc_kernels.sh <tag> "<bench args>" <pass name>:<counters...> [<pass name>:<counters...> ...]
# One rocprofv3 run per pass (at most 8 SQ counters each); output gpurun_out/<tag>/summary.txt:
# per kernel and counter the mean value per launch and the number of launches.
set -u
tag=$1; bargs=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$tag; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for p in "$@"; do
  name=${p%%:*}; ctrs=${p#*:}
  timeout -s KILL 200 rocprofv3 --pmc $ctrs --output-format csv -d $O/$name -- python3 $R/bench.py $bargs --no-cpu-baseline --no-extra > $O/$name.log 2>&1 || { echo "$name failed"; tail -3 $O/$name.log; exit 1; }
done
python3 - $O "$@" > $O/summary.txt <<'PY'
import csv,glob,collections,sys,os
O=sys.argv[1]
for p in sys.argv[2:]:
    d=p.split(":")[0]
    f=sorted(glob.glob(f"{O}/{d}/*/*counter_collection.csv"), key=os.path.getmtime)[-1]
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
        if k.startswith("at::") or "elementwise" in k: continue
        acc[(k,r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k,c),v in sorted(acc.items()): print(f"{d:8s} {k[:60]:60s} {c:28s} launches {len(v):5d} mean {sum(v)/len(v):16.1f} sum {sum(v):18.1f}")
PY
cat $O/summary.txt
