#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r2zz_bench.txt 2>$OUT/r2zz_bench.err; echo rc=$?; tail -2 $OUT/r2zz_bench.err
python -c "
import json
d=json.loads(open('$OUT/r2zz_bench.txt').read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), d['p50_latency_ms']); print(d['cpu_baseline']); print(d['speedup_vs_cpu_baseline']); print(d['result_check'])"
