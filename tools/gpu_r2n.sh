#!/bin/bash
TAG=${1:-r2n}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== bench p2 (host trims)"; timeout 200 python bench.py --steps 10 --warmup 3 --cpu-sample 0 > $OUT/${TAG}_bench.txt 2>$OUT/${TAG}_bench.err; echo "rc=$?"; tail -3 $OUT/${TAG}_bench.err
python - <<PY
import json
d=json.loads(open("$OUT/${TAG}_bench.txt").read().strip().splitlines()[-1])
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "p50", round(d["p50_latency_ms"],3), d["latency_breakdown"])
print(d["roofline"]["isolated"]["stage_ms_per_call"], d["result_check"])
PY
echo "== large case n=100"; timeout 120 python -X faulthandler tools/gpu_large_case.py 100 2>&1 | tail -12
echo "== large case n=200"; timeout 240 python -X faulthandler tools/gpu_large_case.py 200 2>&1 | tail -12
