#!/bin/bash
# Builds lab/libpp_<name>.so with extra -D flags on one source file (experiments only).
# usage: [SRC=pp_rs] tools/variant.sh name -DPP_X=1 ...   (SRC defaults to pp_heuristic)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
src=${SRC:-pp_heuristic}
mkdir -p lab
C=path_planner_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result "$@" -c $C/$src.hip -o lab/${src}_$name.o
objs=$(ls $C/*.o | grep -v /$src.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lab/libpp_$name.so $objs lab/${src}_$name.o
echo lab/libpp_$name.so
