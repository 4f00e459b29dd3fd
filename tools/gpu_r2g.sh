#!/bin/bash
TAG=${1:-r2g}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== overflow probes"; ROMAN_TEST_CAPNNZ=256 timeout 40 python -u tools/gpu_overflow_probe.py small 2>&1 | tail -3; timeout 60 python -u tools/gpu_overflow_probe.py large 2>&1 | tail -3
echo "== all gpu tests"; timeout 480 python -X faulthandler -m pytest tests -q -m gpu -x --durations=6 -o faulthandler_timeout=150 > $OUT/${TAG}_pytest_gpu.txt 2>&1; echo "rc=$?"; grep -v "^  File" $OUT/${TAG}_pytest_gpu.txt | tail -40
for P in 1 2; do
  timeout 150 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --latency-reps 20 --pipeline $P > $OUT/${TAG}_bench_p$P.txt 2>$OUT/${TAG}_bench_p$P.err
  python - $P <<PY
import json,sys
try:
    d=json.loads(open("$OUT/${TAG}_bench_p"+sys.argv[1]+".txt").read().strip().splitlines()[-1])
    print("pipeline", sys.argv[1], "value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "p50", round(d["p50_latency_ms"],3), {k:round(v,3) for k,v in d["stage_ms_per_call"].items()}, "frac", round(d["roofline"]["frac"],3), "iso", d["roofline"].get("isolated",{}).get("per_launch_ms"), d["result_check"])
except Exception as e:
    print("bench failed", e); print(open("$OUT/${TAG}_bench_p"+sys.argv[1]+".err").read()[-2500:])
PY
done
ROMAN_HIP_LIBRARY=$PWD/roman_amd/csrc/variants/libT.so timeout 120 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --latency-reps 2 --pipeline 1 > $OUT/${TAG}_timing.txt 2>$OUT/${TAG}_timing.err
grep -A4 "solve timing" $OUT/${TAG}_timing.err | tail -12
