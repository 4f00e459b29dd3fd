#!/bin/bash
# round-6 odds and ends on ONE box: kernel averages of the final tree (B = 256 launches), then the headline at pipeline depths 2 / 3 / 4.
TAG=${1:-r6n}; export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT
bash tools/gpu_session.sh $TAG prof > /dev/null
python - $OUT/${TAG}_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.9: print(r["Name"][:50], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
for d in 3 4 2 3 4; do
  python bench.py --steps 40 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps -1 --pipeline $d 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('depth $d', round(d['value']), round(d['ms_per_step'], 4))"
done
