#!/bin/bash
# round-6: A/B of environment settings of ONE build on one box, alternating: per setting the rocprofv3 averages of the kernels named
# and the headline.   usage: bash tools/r6_env_ab.sh "k_count<,k_lists" "ROMAN_COUNT_DEAL=0" "ROMAN_COUNT_DEAL=1" [reps]
KERN=$1; A=$2; Bv=$3; REPS=${4:-2}
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
for rep in $(seq $REPS); do for S in "$A" "$Bv"; do
  ( export $S; cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid --check-pairs 0 > /dev/null 2>&1 )
  F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
  python - "$F" "$S" "$KERN" <<'PY'
import csv, sys
o = {}
for r in csv.DictReader(open(sys.argv[1])):
    for k in sys.argv[3].split(','):
        if k in r['Name'] and int(r['Calls']) > 4: o[k] = round(float(r['AverageNs']) / 1e3, 1)
print(sys.argv[2], o)
PY
  rm -rf $OUT/ab_tmp
  ( export $S; timeout 600 python bench.py --steps 40 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps -1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', round(d['value']), round(d['ms_per_step'], 4))" )
done; done
