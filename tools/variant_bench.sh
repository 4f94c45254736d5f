#!/bin/bash
# Times bench workloads under each lab/libpp_*.so variant (GPU box).
# usage: WORKLOADS="analytic queries" tools/variant_bench.sh
set -u
shopt -s nullglob
for so in "" lab/libpp_*.so; do
  for w in ${WORKLOADS:-analytic queries}; do
    PP_HIP_LIB=${so:+$PWD/$so} timeout -k 10 200 python3 -u bench.py --workload $w --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline > /tmp/v.log 2>&1 || { tail -5 /tmp/v.log; exit 1; }
    python3 - /tmp/v.log $w "${so:-default}" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[3], sys.argv[2], "ms", round(d["ms_per_step"],3), d["value"])
PY
  done
done
