#!/bin/bash
# A/B of two library builds on ONE box: per-kernel rocprofv3 averages of bench.py, alternating.   usage: bash tools/ab_libs.sh <libA.so> <libB.so> [kernel-name-substring ...]
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; REPO=$PWD
A=$1; B=$2; shift 2
for rep in 1 2; do for L in $A $B; do
  export ROMAN_HIP_LIBRARY=$REPO/$L
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid > /dev/null 2>&1 )
  F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
  python - "$F" "$L" "$@" <<'PY'
import csv, sys
names = sys.argv[3:] or ['k_cos']
for r in csv.DictReader(open(sys.argv[1])):
    if any(n in r['Name'] for n in names): print(sys.argv[2].split('/')[-1], r['Name'][:34], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
  rm -rf $OUT/ab_tmp
done; done
