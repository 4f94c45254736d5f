#!/bin/bash
# Relaxation band sweep for the heuristic kernel (GPU box).
set -u
O=gpurun_out/wsweep; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_heuristic.py tests/test_tiled_gloo.py -x -q --timeout 180 --timeout-method thread -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for n in 4096 16384; do
  for band in ${BANDS:-8 16 32 64 128}; do
    timeout -k 10 300 python3 -u bench.py --workload wavefront --cells $n --steps 2 --warmup 1 --no-cpu-baseline --band $band > $O/w${n}_$band.log 2>&1 || { tail -5 $O/w${n}_$band.log; exit 1; }
    python3 - "$O/w${n}_$band.log" $n $band <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "band", sys.argv[3], "ms", round(d["ms_per_step"],2), "Gcells/s", round(d["value"]/1e9,2), d["config"]["relaxation"])
PY
  done
done
