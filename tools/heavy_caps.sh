#!/bin/bash
# Search-heavy query leg: how many queries end at the expansion cap, by cap and map density (GPU box).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for spec in ${SPECS:-"0.035:3000" "0.035:8000" "0.035:20000"}; do
  dens=${spec%%:*}; cap=${spec#*:}
  timeout -k 10 200 python3 bench.py --workload queries --heavy --heavy-density $dens --steps 1 --warmup 1 --no-cpu-baseline --max-expansions $cap > /tmp/h.log 2>&1 || { tail -3 /tmp/h.log; exit 1; }
  python3 - $dens $cap <<'PY'
import json,sys
d=json.loads(open('/tmp/h.log').read().strip().splitlines()[-1])
s=d['search']; print("density",sys.argv[1],"cap",sys.argv[2],"ms",round(d['ms_per_step'],1),"found",s['found'],"no_path",s['no_path'],"cap_reached",s['cap_reached'],"exp",s['expansions'],"median",s['expansions_median'],"rounds",s['per_rank'][0]['rounds'])
PY
done
