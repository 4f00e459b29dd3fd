#!/bin/bash
# One GPU-box session: GPU tests, smoke, bench, rocprofv3 kernel stats.  Outputs under gpurun_out/.
# usage (from the repo root on the box): bash tools/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu" ; timeout 1500 python -m pytest tests -q -m gpu > $OUT/${TAG}_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 $OUT/${TAG}_pytest_gpu.txt
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $OUT/${TAG}_smoke.txt
echo "== bench" ; timeout 900 python bench.py --steps 10 --warmup 2 > $OUT/${TAG}_bench.txt 2>$OUT/${TAG}_bench.err; echo "bench rc=$?"; tail -1 $OUT/${TAG}_bench.txt; tail -3 $OUT/${TAG}_bench.err
echo "== rocprofv3 kernel stats"
REPO=$PWD
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $REPO/bench.py --steps 5 --warmup 1 --cpu-sample 0 --latency-reps 5 > $OUT/${TAG}_prof_bench.txt 2>$OUT/${TAG}_prof.err ); echo "rocprof rc=$?"
find $OUT/${TAG}_prof -name "*stats*" | head; 
F=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -20 "$F"
# keep the merge small: drop the big per-dispatch trace
find $OUT/${TAG}_prof -name "*kernel_trace.csv" -size +20M -delete
