#!/bin/bash
# Final session of round 2: all GPU tests, smoke, the bench as the driver runs it, rocprofv3 kernel stats of the
# same bench at one batch in flight (isolated kernels).  Outputs under gpurun_out/.
TAG=${1:-r2final}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$PWD
echo "== pytest -m gpu"; timeout 500 python -X faulthandler -m pytest -o faulthandler_timeout=200 tests -q -m gpu --durations=5 > $OUT/${TAG}_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -9 $OUT/${TAG}_pytest_gpu.txt
echo "== smoke"; timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $OUT/${TAG}_smoke.txt
echo "== bench (driver flags)"; timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench.txt 2>$OUT/${TAG}_bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$OUT/${TAG}_bench.txt").read().strip().splitlines()[-1])
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "p50", round(d["p50_latency_ms"],3))
print(d["roofline"]); print(d["cpu_baseline"]); print(d["result_check"])
PY
echo "== rocprofv3 kernel stats (pipeline 1)"
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --cpu-sample 0 --latency-reps -1 --pipeline 1 > $OUT/${TAG}_prof_bench.txt 2>$OUT/${TAG}_prof.err ); echo "rocprof rc=$?"
F=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -24 "$F"
find $OUT/${TAG}_prof -name "*kernel_trace.csv" -size +20M -delete
