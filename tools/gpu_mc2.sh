#!/bin/bash
for cfg in "2 0" "2 4" "2 8" "1 4" "1 3" "3 8"; do
  set -- $cfg
  ROMAN_MAX_COMPACT=$1 ROMAN_LATE_RATIO=$2 timeout 300 python bench.py --steps 12 --warmup 3 --cpu-sample 0 --latency-reps -1 --pipeline 1 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('budget $1 late_ratio $2 value', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'solve', round(d['stage_ms_per_step']['solve'],3))"
done
for cfg in "0 0" "0 4" "1 0"; do
  set -- $cfg
  ROMAN_MAX_COMPACT=$1 ROMAN_LATE_RATIO=$2 timeout 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --latency-reps 40 --pipeline 1 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=1 budget $1 late_ratio $2 p50', round(d['p50_latency_ms'],4))"
done
