#!/bin/bash
TAG=${1:-r2u}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 150 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --latency-reps 5 --also-grid > $OUT/${TAG}_bench.txt 2>$OUT/${TAG}_bench.err; echo "rc=$?"; tail -3 $OUT/${TAG}_bench.err
python - <<PY
import json
d=json.loads(open("$OUT/${TAG}_bench.txt").read().strip().splitlines()[-1])
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "p50", round(d["p50_latency_ms"],3), d["scaling"], d["config"]["workload"][:60])
print(d["grid_config4"])
PY
