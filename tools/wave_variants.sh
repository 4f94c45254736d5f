#!/bin/bash
# tools/wave_bench.sh under every lab/libpp_*.so variant (GPU box).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
shopt -s nullglob
for so in lab/libpp_*.so; do
  echo "== $so"
  LIB=$R/$so bash tools/wave_bench.sh || exit 1
done
echo "== default"
bash tools/wave_bench.sh
