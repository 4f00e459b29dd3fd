#!/bin/bash
TAG=${1:-r2p}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== bench p2"; timeout 200 python bench.py --steps 10 --warmup 3 --cpu-sample 0 > $OUT/${TAG}_bench.txt 2>$OUT/${TAG}_bench.err; echo "rc=$?"; tail -3 $OUT/${TAG}_bench.err
python - <<PY
import json
d=json.loads(open("$OUT/${TAG}_bench.txt").read().strip().splitlines()[-1])
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "p50", round(d["p50_latency_ms"],3), d["latency_breakdown"])
print(d["roofline"]["isolated"]["stage_ms_per_call"], d["result_check"])
PY
echo "== SPI=16 for comparison"; ROMAN_FILL_SPI=16 timeout 200 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --latency-reps 2 > $OUT/${TAG}_bench16.txt 2>/dev/null
python - <<PY
import json
d=json.loads(open("$OUT/${TAG}_bench16.txt").read().strip().splitlines()[-1])
print("SPI16 value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), d["roofline"]["isolated"]["stage_ms_per_call"])
PY
echo "== parity subset"; timeout 300 python -X faulthandler -m pytest -o faulthandler_timeout=120 tests/test_gpu_parity.py tests/test_gpu_golden.py -q -x -m gpu 2>&1 | tail -4
