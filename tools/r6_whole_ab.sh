#!/bin/bash
# round-6: k_count's work items — whole problems (default for batches of >= one problem per compute unit) against blocks of 128 rows
# (ROMAN_COUNT_WHOLE=0) on ONE box, alternating: kernel averages of k_count and of k_lists, which reads its mask rows, and the headline.
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
for wv in 1 0 1 0; do
  export ROMAN_COUNT_WHOLE=$wv
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid --check-pairs 0 > /dev/null 2>&1 )
  F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
  python - "$F" "whole=$wv" <<'PY'
import csv, sys
o = {}
for r in csv.DictReader(open(sys.argv[1])):
    for k in ('k_count<', 'k_lists', 'k_solve_up<8', 'k_fill_list'):
        if k in r['Name'] and int(r['Calls']) > 4: o[k] = round(float(r['AverageNs']) / 1e3, 1)
print(sys.argv[2], o)
PY
  rm -rf $OUT/ab_tmp
  timeout 600 python bench.py --steps 40 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps -1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', round(d['value']), round(d['ms_per_step'], 4))"
done
