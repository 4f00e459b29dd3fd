#!/bin/bash
# round-6: the stream solver's bounded launches (ROMAN_SOLVE_CAP: default = 64 passes when a call holds more problems than workgroups,
# 0 = never) on ONE box: the bit-for-bit test, the full-config parity tests, then per setting the kernel averages of an isolated launch,
# the headline, and the caller legs (one shot of the 4096-pair grid, the 8-GPU projection) from the bench's full record.
TAG=${1:-r6m}; export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 1200 python -m pytest tests -q -x -m gpu -k "bounded_solver or config3 or config4 or ragged or two_batches or candidate" > $OUT/${TAG}_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $OUT/${TAG}_pytest.txt
for cap in default 0 default 0; do
  if [ "$cap" = "default" ]; then unset ROMAN_SOLVE_CAP; else export ROMAN_SOLVE_CAP=$cap; fi
  timeout 900 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --check-pairs 0 > $OUT/${TAG}_bench_cap$cap.txt 2>/dev/null
  python - $OUT/${TAG}_bench_cap$cap.txt bench_extras.json "cap=$cap" <<'PY'
import json, sys
line = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); ex = json.load(open(sys.argv[2]))
c = ex.get("caller", {})
one = c.get("one_shot", {}); proj = c.get("scale_projection", {})
print(sys.argv[3], "value", round(line["value"]), "grid", round((line.get("grid_config4") or {}).get("value", 0)), "p50", line.get("p50_latency_ms"),
      "| one_shot", [(r["chunk"], round(r["ms"], 2), round(r["vs_steady_state"], 3)) for r in one.get("rows", [])],
      "| projection", [(r["chunk"], round(r["min_projection"], 2), [round(x, 2) for x in r["rank_ms"]]) for r in proj.get("rows", [])])
PY
  cp bench_extras.json $OUT/${TAG}_extras_cap$cap.json
done
