#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 80 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 > $OUT/r2z_bench.txt 2>/dev/null
python -c "
import json
d=json.loads(open('$OUT/r2z_bench.txt').read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), d['p50_latency_ms'], d['config']['batches_in_flight'], d['roofline']['frac'], d['roofline']['isolated']['frac'])"
