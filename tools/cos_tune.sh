#!/bin/bash
# experiment: k_cos_tile with parts switched off (ROMAN_COS_TUNE bits: 1 no MFMA, 2 no global loads after stage 0, 4 no LDS stores, 8 no barriers)
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; REPO=$PWD
for kc in 16; do for t in 14 30 46 62 126; do
  export ROMAN_COS=$kc ROMAN_COS_TUNE=$t
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ct_${kc}_$t -o b -- python $REPO/bench.py --steps 3 --warmup 1 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid > /dev/null 2>&1 )
  F=$(find $OUT/ct_${kc}_$t -name "*kernel_stats.csv" | head -1)
  python - "$F" "KC=$kc tune=$t" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_cos' in r['Name']: print(sys.argv[2], r['Name'][:30], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
  rm -rf $OUT/ct_${kc}_$t
done; done
