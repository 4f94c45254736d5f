#!/bin/bash
# Host-side AddressSanitizer run of the planner (device code is compiled as usual).
#   build (here, no GPU needed):  tools/asan_planner.sh build     -> lab/libpp_asan.so, lab/asan_planner
#   run (GPU box):                tools/asan_planner.sh run [queries] [expansion cap]
set -eu
cd "$(dirname "$0")/.."
C=path_planner_amd/csrc
case ${1:-build} in
  build)
    mkdir -p lab
    for f in pp_astar pp_collision pp_ctx pp_expand pp_heuristic pp_planner pp_rs; do
      /opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result \
        -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -c $C/$f.hip -o lab/asan_$f.o &
    done
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-sanitize -fsanitize=address -o lab/libpp_asan.so lab/asan_*.o
    /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fno-gpu-sanitize -fsanitize=address -fno-omit-frame-pointer -I include tools/asan_planner.cpp \
      -o lab/asan_planner -L lab -lpp_asan -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib
    rm -f lab/asan_*.o
    ;;
  run)
    ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 timeout -k 10 300 lab/asan_planner ${2:-1100} ${3:-150}
    ;;
esac
