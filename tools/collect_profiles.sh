#!/bin/bash
# Copies the newest outputs of tools/gpu_measure.sh (gpurun_out/measure) into profiles/.
set -e
cd "$(dirname "$0")/.."
M=gpurun_out/measure
for p in collision:collision rs:rs wave:wavefront16384 queries:queries; do
  src=${p%%:*}; dst=${p##*:}
  f=$(ls -t $M/prof_$src/vm/*_kernel_stats.csv | head -1)
  cp "$f" profiles/r01_bench_${dst}_kernel_stats.csv
done
for w in collision rs wave4096 wave16384 expand analytic queries; do tail -1 $M/bench_$w.log > profiles/r01_bench_$w.json; done
python3 - <<'PY'
import json
for w in ("collision","rs","wave4096","wave16384","expand","analytic","queries"):
    d=json.loads(open(f"profiles/r01_bench_{w}.json").read())
    r=d.get("roofline") or {}
    print(w.ljust(10), d["metric"].ljust(36), f'{d["value"]:.4g}'.rjust(10), d["unit"].ljust(13), f'{d["ms_per_step"]:.3f} ms', "frac", r.get("frac"), "| cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
