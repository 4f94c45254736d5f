#!/bin/bash
# Runs the given pytest selection under the default library and each lab/libpp_*.so (GPU box).
# usage: tools/variant_test.sh tests/test_gpu_rs.py -k analytic
set -u
shopt -s nullglob
for so in "" lab/libpp_*.so; do
  echo "== ${so:-default}"
  PP_HIP_LIB=${so:+$PWD/$so} timeout -k 10 300 python3 -u -m pytest "$@" -x -q --timeout 120 --timeout-method thread -m gpu > /tmp/vt.log 2>&1 || { tail -30 /tmp/vt.log; exit 1; }
  tail -1 /tmp/vt.log
done
