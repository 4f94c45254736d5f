#!/bin/bash
# Times the wavefront bench under each lab/libpp_*.so variant (GPU box).
set -u
shopt -s nullglob
for so in "" lab/libpp_*.so; do
  for n in ${SIZES:-4096 16384}; do
    PP_HIP_LIB=${so:+$PWD/$so} timeout -k 10 200 python3 -u bench.py --workload wavefront --cells $n --steps 2 --warmup 1 --no-cpu-baseline --band ${BAND:-128} > /tmp/v.log 2>&1 || { tail -5 /tmp/v.log; exit 1; }
    python3 - /tmp/v.log $n "${so:-default}" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[3], sys.argv[2], "ms", round(d["ms_per_step"],2), d["config"]["relaxation"])
PY
  done
done
