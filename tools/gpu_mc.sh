#!/bin/bash
# batch throughput for several compaction budgets.  usage: bash tools/gpu_mc.sh "budgets" pipeline
for mc in $1; do
  ROMAN_MAX_COMPACT=$mc timeout 300 python bench.py --steps 12 --warmup 3 --cpu-sample 0 --latency-reps -1 --pipeline ${2:-1} 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('max_compact $mc pipeline ${2:-1} value', round(d['value']), 'ms/step', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d.get('stage_ms_per_step',{}).items()})"
done
