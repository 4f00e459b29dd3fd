#!/bin/bash
TAG=${1:-r2t}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== bench"; timeout 200 python bench.py --steps 10 --warmup 3 --cpu-sample 0 > $OUT/${TAG}_bench.txt 2>$OUT/${TAG}_bench.err; echo "rc=$?"; tail -3 $OUT/${TAG}_bench.err
python - <<PY
import json
d=json.loads(open("$OUT/${TAG}_bench.txt").read().strip().splitlines()[-1])
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "p50", round(d["p50_latency_ms"],3), d["latency_breakdown"])
print(d["roofline"]["isolated"]["stage_ms_per_call"], d["result_check"])
PY
echo "== tests"; timeout 300 python -X faulthandler -m pytest -o faulthandler_timeout=100 tests/test_gpu_batch.py tests/test_gpu_shim.py tests/test_gpu_full_configs.py -q -x -m gpu -k "not config3 and not config4" 2>&1 | tail -6
