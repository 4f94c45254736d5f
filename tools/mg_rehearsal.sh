#!/bin/bash
# Two ranks sharing the box's one GPU over gloo: every bench workload through the multi-rank path (GPU box).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for w in "collision --poses 4194304" "rs --pairs 262144" "queries --queries 1024" "wavefront --cells 4096" "expand --parents 65536" "analytic --pairs 65536"; do
  PP_BENCH_BACKEND=gloo timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 2 --workload $w --steps 3 --warmup 1 > /tmp/mg.log 2>&1 || { tail -20 /tmp/mg.log; exit 1; }
  tail -n 1 /tmp/mg.log | cut -c1-330
done
