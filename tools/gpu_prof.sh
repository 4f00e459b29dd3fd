#!/bin/bash
# rocprofv3 kernel stats of the bench workload.  usage: bash tools/gpu_prof.sh [tag]
TAG=${1:-p}
OUT=$PWD/gpurun_out
REPO=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $REPO/bench.py --steps 5 --warmup 1 --cpu-sample 0 --latency-reps 5 $ROMAN_BENCH_ARGS > $OUT/${TAG}_prof_bench.txt 2>$OUT/${TAG}_prof.err ); echo "rocprof rc=$?"
F=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cut -c1-60,200- "$F" | head -16
[ -n "$F" ] && python3 - "$F" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:50]:50s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} max_us={float(r['MaxNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
find $OUT/${TAG}_prof -name "*kernel_trace.csv" -size +20M -delete
