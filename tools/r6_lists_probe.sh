#!/bin/bash
# round-6 probe: k_lists' duration per PROCESS behind the prefiltered k_count — whole problems (default) against row blocks —, the same
# command several times on one box (is the slow mode a property of the process, the box or the setting?).
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
for S in "ROMAN_COUNT_WHOLE=1" "ROMAN_COUNT_WHOLE=1" "ROMAN_COUNT_WHOLE=0" "ROMAN_COUNT_WHOLE=1" "ROMAN_COUNT_WHOLE=1 ROMAN_LISTS_LDS=30000" "ROMAN_COUNT_WHOLE=1 ROMAN_LISTS_LDS=30000" "ROMAN_COUNT_WHOLE=0" "ROMAN_COUNT_WHOLE=1" "ROMAN_COUNT_WHOLE=1 ROMAN_COUNT_PRE=0"; do
  ( export $S; cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid --check-pairs 0 > /dev/null 2>&1 )
  F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
  python - "$F" "$S" <<'PY'
import csv, sys
o = {}
for r in csv.DictReader(open(sys.argv[1])):
    for k in ('k_count<', 'k_lists', 'k_fill_list', 'k_solve_up<8'):
        if k in r['Name'] and int(r['Calls']) > 4: o[k] = round(float(r['AverageNs']) / 1e3, 1)
print(sys.argv[2].ljust(48), o)
PY
  rm -rf $OUT/ab_tmp
done
