#!/bin/bash
# Hardware counters of the collision kernel (GPU box): one pass per counter group.
# usage: [PP_COLLISION_KERNEL=queue|lane] tools/k2_counters.sh [tag]
set -u
tag=${1:-k2pmc}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$tag; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $O/$name -- python3 $R/bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 > $O/$name.log 2>&1 || { echo "$name failed"; tail -3 $O/$name.log; exit 1; }; }
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE GRBM_COUNT
run tcp TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_CVT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_WAIT_ANY
python3 - $O <<'PY'
import csv,glob,collections,sys
O=sys.argv[1]
for d in ("sq","tcp","sq2"):
    f=glob.glob(f"{O}/{d}/*/*counter_collection.csv")[0]
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_collision" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(d,k,len(v),sum(v)/len(v))
PY
