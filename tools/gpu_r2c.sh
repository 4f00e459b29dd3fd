#!/bin/bash
# round-2 session c: first run of the upper-triangle stream solver
TAG=${1:-r2c}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== quick parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu > $OUT/${TAG}_parity.txt 2>&1; echo "rc=$?"; tail -25 $OUT/${TAG}_parity.txt
echo "== all gpu tests"; timeout 2400 python -m pytest tests -q -m gpu --durations=5 > $OUT/${TAG}_pytest_gpu.txt 2>&1; echo "rc=$?"; tail -40 $OUT/${TAG}_pytest_gpu.txt
for P in 1 2; do
  timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --latency-reps 20 --pipeline $P > $OUT/${TAG}_bench_p$P.txt 2>$OUT/${TAG}_bench_p$P.err
  python - $P <<PY
import json,sys
try:
    d=json.loads(open("$OUT/${TAG}_bench_p"+sys.argv[1]+".txt").read().strip().splitlines()[-1])
    print("pipeline", sys.argv[1], "value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "p50", round(d["p50_latency_ms"],3), {k:round(v,3) for k,v in d["stage_ms_per_step"].items()}, "frac", round(d["roofline"]["frac"],3), d["result_check"])
except Exception as e:
    print("bench failed", e); print(open("$OUT/${TAG}_bench_p"+sys.argv[1]+".err").read()[-2500:])
PY
done
ROMAN_HIP_LIBRARY=$PWD/roman_amd/csrc/variants/libT.so timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --latency-reps 2 --pipeline 1 > $OUT/${TAG}_timing.txt 2>$OUT/${TAG}_timing.err
grep -A4 "solve timing" $OUT/${TAG}_timing.err | tail -12
