#!/bin/bash
TAG=${1:-r2k}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$PWD
echo "== fill timing"; ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/libF.so timeout 100 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --latency-reps 1 --pipeline 1 > $OUT/${TAG}_ftiming.txt 2>$OUT/${TAG}_ftiming.err; grep "fill timing" $OUT/${TAG}_ftiming.err | sort | uniq -c | sort -rn | head -6
echo "== solve timing"; ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/libT.so timeout 100 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --latency-reps 2 --pipeline 1 > $OUT/${TAG}_stiming.txt 2>$OUT/${TAG}_stiming.err; grep -A4 "solve timing" $OUT/${TAG}_stiming.err | tail -12
echo "== bench p2"; timeout 200 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --check-pairs 0 > $OUT/${TAG}_bench.txt 2>$OUT/${TAG}_bench.err; echo "rc=$?"
python - <<PY
import json
d=json.loads(open("$OUT/${TAG}_bench.txt").read().strip().splitlines()[-1])
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "p50", round(d["p50_latency_ms"],3), d["latency_breakdown"])
print(d["roofline"]["isolated"]["stage_ms_per_call"])
PY
