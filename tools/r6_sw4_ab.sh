#!/bin/bash
# round-6: the stream solver with four waves per problem (variants/libSW4.so: two problems per compute unit) against eight, ONE box:
# solver-facing tests on the variant, then kernel averages, headline, grid and p50 for both, alternating.
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/libSW4.so timeout 900 python -m pytest tests -q -m gpu -k "stagewise or config3 or ragged or fixed_point or tie_fallback or explicit_u0 or u0_stability" 2>&1 | tail -6
for L in SW4 default SW4 default; do
  if [ "$L" = "default" ]; then unset ROMAN_HIP_LIBRARY; else export ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/lib$L.so; fi
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid --check-pairs 0 > /dev/null 2>&1 )
  F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
  python - "$F" "$L" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_solve_up<' in r['Name'] and int(r['Calls']) > 4: print(sys.argv[2], r['Name'][:44], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
  rm -rf $OUT/ab_tmp
  timeout 600 python bench.py --steps 40 --warmup 5 --no-extras --cpu-sample 0 --check-pairs 0 --latency-reps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   $L value', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'grid', round((d.get('grid_config4') or {}).get('value',0)), 'p50', d['p50_latency_ms'], 'iso_ms', d['roofline']['isolated']['avg_launch_ms'])"
done
