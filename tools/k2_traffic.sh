#!/bin/bash
# HBM-side traffic counters of the collision kernel (GPU box); usage: tools/k2_traffic.sh tag
set -u
tag=${1:-k2traffic}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$tag; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $O/$name -- python3 $R/bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 > $O/$name.log 2>&1 || { echo "$name failed"; tail -3 $O/$name.log; exit 1; }; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_WRITE_sum
run tcc2 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
python3 - $O <<'PY'
import csv,glob,collections,sys
O=sys.argv[1]
for d in ("fetch","write","tcc","tcc2"):
    f=glob.glob(f"{O}/{d}/*/*counter_collection.csv")[0]
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_collision" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(d,k,len(v),sum(v)/len(v))
PY
