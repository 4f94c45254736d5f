#!/bin/bash
# Two ranks sharing the box's one GPU over gloo: the slab wavefront at 4096^2 for several bands (GPU box).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for band in ${BANDS:-256 1024 1e9}; do
  PP_BENCH_BACKEND=gloo timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --workload wavefront --cells ${CELLS:-4096} --steps 3 --warmup 1 --slab-band $band > /tmp/mg.log 2>&1 || { tail -20 /tmp/mg.log; exit 1; }
  echo "band $band: $(grep -o '"ms_per_step": [0-9.]*' /tmp/mg.log) $(grep -o '"exchange_rounds": [0-9]*' /tmp/mg.log) $(grep -o '"relaxation": {[^}]*}' /tmp/mg.log)"
done
