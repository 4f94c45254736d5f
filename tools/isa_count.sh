#!/bin/bash
# usage: tools/isa_count.sh <file.hip> <kernel-substring>  -- VALU/VMEM/SALU instruction counts of one kernel
f=$1; k=$2
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I path_planner_amd/csrc --cuda-device-only -S $f -o /tmp/isa.s 2>/dev/null
awk -v k="$k" 'index($0, k) && /^_Z[^ ]*:/ {on=1} on && /s_endpgm/ {on=0} on {print}' /tmp/isa.s > /tmp/isa_k.s
echo "total lines: $(wc -l < /tmp/isa_k.s)"
echo "VALU: $(grep -cE '^\s+v_' /tmp/isa_k.s)  f64: $(grep -cE '^\s+v_[a-z0-9_]*f64' /tmp/isa_k.s)  VMEM: $(grep -cE '^\s+(global|buffer|flat|scratch)_' /tmp/isa_k.s)  SALU: $(grep -cE '^\s+s_' /tmp/isa_k.s) LDS: $(grep -cE '^\s+ds_' /tmp/isa_k.s)"
grep -E "vgpr_count|sgpr_count|scratch" /tmp/isa.s | head -8
