#!/bin/bash
# Instruction and stall counters of the collision kernel for the demo and the compact car (GPU box).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
for car in demo compact; do
  bash $R/tools/pmc_kernels.sh pmc_k2_$car "--workload collision --car $car --steps 3 --warmup 1" "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "sq2:SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY" > /dev/null || exit 1
  echo "== $car"; python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_k2_$car sq sq2 | grep k_collision
done
