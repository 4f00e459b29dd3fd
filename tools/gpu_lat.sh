#!/bin/bash
# p50 single-pair latency for several compaction budgets.  usage: bash tools/gpu_lat.sh
for mc in 0 1 2 1000; do
  ROMAN_MAX_COMPACT=$mc timeout 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --latency-reps 40 --pipeline 1 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('max_compact $mc p50', round(d['p50_latency_ms'],4), 'ms/step', round(d['ms_per_step'],3))"
done
