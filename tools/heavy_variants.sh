#!/bin/bash
# The search-heavy and the light queries batch under every lab/libpp_*.so and the default library (GPU box).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
shopt -s nullglob
for so in "" lab/libpp_*.so; do
  for w in "--heavy --steps 2" "--steps 4"; do
    PP_PLAN_PROFILE=1 PP_HIP_LIB=${so:+$R/$so} timeout -k 10 300 python3 -u bench.py --workload queries $w --warmup 1 --no-cpu-baseline > /tmp/hv.log 2>&1 || { tail -5 /tmp/hv.log; exit 1; }
    echo "${so:-default} [$w] $(tail -n 1 /tmp/hv.log | grep -o '"ms_per_step": [0-9.]*') $(grep 'large rounds' /tmp/hv.log | tail -n 1 | cut -d'|' -f2)"
  done
done
