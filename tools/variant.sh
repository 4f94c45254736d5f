#!/bin/bash
# Builds lab/libpp_<name>.so with extra -D flags on pp_heuristic.hip (experiments only).
# usage: tools/variant.sh name -DPP_X=1 ...
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p lab
C=path_planner_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result "$@" -c $C/pp_heuristic.hip -o lab/pp_heuristic_$name.o
objs=$(ls $C/*.o | grep -v pp_heuristic.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lab/libpp_$name.so $objs lab/pp_heuristic_$name.o
echo lab/libpp_$name.so
