#!/bin/bash
TAG=${1:-r2s}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for ng in 0 2 4 6; do
if [ $ng = 0 ]; then unset ROMAN_FILL_NG; else export ROMAN_FILL_NG=$ng; fi
timeout 100 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --latency-reps 2 > $OUT/${TAG}_b$ng.txt 2>/dev/null
python - <<PY
import json
d=json.loads(open("$OUT/${TAG}_b$ng.txt").read().strip().splitlines()[-1])
print("NG $ng value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), d["roofline"]["isolated"]["stage_ms_per_call"])
PY
done
unset ROMAN_FILL_NG
echo "== grid"; timeout 200 python bench.py --workload grid --steps 2 --warmup 1 --cpu-sample 0 --latency-reps 2 > $OUT/${TAG}_grid.txt 2>/dev/null
python - <<PY
import json
d=json.loads(open("$OUT/${TAG}_grid.txt").read().strip().splitlines()[-1])
print("grid value", round(d["value"]), "ms/step", round(d["ms_per_step"],2), d["result_check"])
PY
echo "== tests"; timeout 300 python -X faulthandler -m pytest -o faulthandler_timeout=120 tests/test_gpu_parity.py tests/test_gpu_batch.py -q -x -m gpu 2>&1 | tail -4
