#!/bin/bash
# round-6 probe 2: k_lists behind the prefiltered k_count whose mask rows are written with non-temporal stores (variants/libNT.so).
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
for L in default NT default NT; do
  if [ "$L" = "default" ]; then unset ROMAN_HIP_LIBRARY; else export ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/lib$L.so; fi
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid --check-pairs 0 > /dev/null 2>&1 )
  F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
  python - "$F" "$L" <<'PY'
import csv, sys
o = {}
for r in csv.DictReader(open(sys.argv[1])):
    for k in ('k_count<', 'k_lists', 'k_fill_list', 'k_solve_up<8'):
        if k in r['Name'] and int(r['Calls']) > 4: o[k] = round(float(r['AverageNs']) / 1e3, 1)
print(sys.argv[2].ljust(10), o)
PY
  rm -rf $OUT/ab_tmp
  timeout 600 python bench.py --steps 40 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps -1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', round(d['value']), round(d['ms_per_step'], 4))"
done
