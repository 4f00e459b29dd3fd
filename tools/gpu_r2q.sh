#!/bin/bash
TAG=${1:-r2q}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for spi in 3 4 6 8 11 12; do
ROMAN_FILL_SPI=$spi timeout 100 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --latency-reps 2 > $OUT/${TAG}_b$spi.txt 2>/dev/null
python - <<PY
import json
d=json.loads(open("$OUT/${TAG}_b$spi.txt").read().strip().splitlines()[-1])
print("SPI $spi value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), d["roofline"]["isolated"]["stage_ms_per_call"])
PY
done
