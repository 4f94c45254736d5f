#!/bin/bash
# fp64 / LDS / occupancy counter passes for the Reeds-Shepp, analytic, rollout and heuristic kernels,
# plus the heuristic's HBM traffic (GPU box).  Summaries land in gpurun_out/pmc_<workload>/summary.txt.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
F64="f64:SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES"
LDS="lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES"
OCC="occ:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
for w in ${WORKLOADS:-rs analytic expand wavefront}; do
  case $w in
    rs|analytic) bash $R/tools/pmc_kernels.sh pmc_$w "--workload $w --steps 1 --warmup 1" "$F64" "$LDS" "$OCC" || exit 1;;
    expand) bash $R/tools/pmc_kernels.sh pmc_$w "--workload $w --steps 1 --warmup 1" "$F64" "$OCC" || exit 1;;
    wavefront) bash $R/tools/pmc_kernels.sh pmc_$w "--workload wavefront --cells 16384 --steps 1 --warmup 0" "$F64" "$LDS" "$OCC" "fetch:FETCH_SIZE" "write:WRITE_SIZE" || exit 1;;
  esac
done
