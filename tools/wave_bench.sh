#!/bin/bash
# Heuristic relaxation timings (GPU box): 4096^2, 16384^2 and the walls map, optional band sweep.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for spec in ${SPECS:-4096:bernoulli 16384:bernoulli 4096:walls}; do  # SPECS="16384:bernoulli 4096:walls"
  set -- ${spec/:/ }
  for band in ${BANDS:-128}; do
    PP_HIP_LIB=${LIB:-} timeout -k 10 300 python3 -u bench.py --workload wavefront --cells $1 --map-kind $2 --steps 3 --warmup 1 --no-cpu-baseline --band $band > /tmp/w.log 2>&1 || { tail -5 /tmp/w.log; exit 1; }
    python3 - /tmp/w.log $1 $2 $band <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], sys.argv[3], "band", sys.argv[4], "ms", round(d["ms_per_step"],2), d["config"]["relaxation"])
PY
  done
done
