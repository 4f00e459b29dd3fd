#!/bin/bash
# Quick GPU check: GPU tests (stop at first failure) + bench.  usage: bash tools/gpu_quick.sh [tag] [pytest -k expr]
TAG=${1:-q}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -x -m gpu ${2:+-k "$2"} > $OUT/${TAG}_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -15 $OUT/${TAG}_pytest_gpu.txt
timeout 600 python bench.py --steps 10 --warmup 2 > $OUT/${TAG}_bench.txt 2>$OUT/${TAG}_bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d=json.loads(open("$OUT/${TAG}_bench.txt").read().strip().splitlines()[-1])
    print("value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "p50", round(d["p50_latency_ms"],3), "stages", {k:round(v,3) for k,v in d["stage_ms_per_step"].items()}, "frac", round(d["roofline"]["frac"],3), "same", d.get("cpu_baseline",{}).get("identical_to_gpu"))
except Exception as e:
    print("bench parse failed", e); print(open("$OUT/${TAG}_bench.err").read()[-2000:])
PY
