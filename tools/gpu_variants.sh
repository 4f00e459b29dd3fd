#!/bin/bash
# Time bench.py stages with alternative builds of the library (timing experiments only).
OUT=$PWD/gpurun_out; mkdir -p $OUT
for f in "$@"; do
  ROMAN_HIP_LIBRARY=$PWD/$f timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --latency-reps 3 --pipeline ${PIPE:-1} > $OUT/var.txt 2>$OUT/var.err
  python - "$f" <<PY
import json,sys
try:
    d=json.loads(open("$OUT/var.txt").read().strip().splitlines()[-1])
    print(sys.argv[1], "ms/step", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["stage_ms_per_step"].items()}, "recall", d["result_check"]["planted_inlier_recall_mean"])
except Exception as e:
    print(sys.argv[1], "failed", e); print(open("$OUT/var.err").read()[-1500:])
PY
done
