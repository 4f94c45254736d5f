#!/bin/bash
# Times the collision bench under the default library and each lab/libpp_*.so (built by tools/variant.sh).
set -u
shopt -s nullglob
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for so in "" lab/libpp_*.so; do
  for k in bench; do
    PP_HIP_LIB=${so:+$PWD/$so} timeout -k 10 200 python3 -u bench.py --no-cpu-baseline > /tmp/v.log 2>&1 || { tail -5 /tmp/v.log; exit 1; }
    python3 - /tmp/v.log $k "${so:-default}" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[3], sys.argv[2], "kernel_ms", round(d["roofline"]["kernel_ms"],4), "frac", round(d["roofline"]["frac"],3))
PY
  done
done
