#!/bin/bash
# rocprofv3 kernel stats of the bench at its default depth (two batches in flight), batch launches only
TAG=${1:-r2x}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$PWD
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $REPO/bench.py --steps 20 --warmup 5 --cpu-sample 0 --latency-reps -1 > $OUT/${TAG}_prof_bench.txt 2>$OUT/${TAG}_prof.err ); echo "rocprof rc=$?"
F=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -8 "$F" | cut -c1-200
python - <<PY
import json
d=json.loads(open("$OUT/${TAG}_prof_bench.txt").read().strip().splitlines()[-1])
print("value under rocprof", round(d["value"]), "solve avg ms (hipEvents)", d["roofline"]["avg_launch_ms"])
PY
find $OUT/${TAG}_prof -name "*kernel_trace.csv" -size +20M -delete
