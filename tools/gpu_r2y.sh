#!/bin/bash
export TMPDIR=/tmp
for p in 3 2; do
timeout 60 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --latency-reps -1 --pipeline $p 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipeline $p', round(d['value']), round(d['ms_per_step'],3))"
done
