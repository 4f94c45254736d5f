#!/bin/bash
# HBM traffic of one 16384^2 heuristic relaxation, per kernel (GPU box): FETCH_SIZE and WRITE_SIZE passes.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/pmc_kernels.sh pmc_wavefront "--workload wavefront --cells 16384 --steps 1 --warmup 0" "fetch:FETCH_SIZE" "write:WRITE_SIZE" > /dev/null || exit 1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_wavefront fetch write
