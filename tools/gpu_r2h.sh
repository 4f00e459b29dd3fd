#!/bin/bash
TAG=${1:-r2h}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== overflow probes"; ROMAN_TEST_CAPNNZ=256 timeout 40 python -u tools/gpu_overflow_probe.py small 2>&1 | tail -1
echo "== all gpu tests"; timeout 420 python -X faulthandler -m pytest tests -q -m gpu -x --durations=6 -o faulthandler_timeout=100 > $OUT/${TAG}_pytest_gpu.txt 2>&1; echo "rc=$?"; grep -v "^  File" $OUT/${TAG}_pytest_gpu.txt | tail -30
bench() {  # tag, extra env/lib
  timeout 150 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --latency-reps 20 --pipeline $2 > $OUT/${TAG}_bench_$1.txt 2>$OUT/${TAG}_bench_$1.err
  python - $1 <<PY
import json,sys
try:
    d=json.loads(open("$OUT/${TAG}_bench_"+sys.argv[1]+".txt").read().strip().splitlines()[-1])
    print(sys.argv[1], "value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "p50", round(d["p50_latency_ms"],3), {k:round(v,3) for k,v in d["stage_ms_per_call"].items()}, "iso", [round(x,3) for x in d["roofline"].get("isolated",{}).get("per_launch_ms",[])], "iso stages", {k:round(v,3) for k,v in d["roofline"].get("isolated",{}).get("stage_ms_per_call",{}).items()}, d["result_check"]["status_ok_frac"], d["result_check"]["planted_inlier_recall_mean"])
except Exception as e:
    print("bench failed", e); print(open("$OUT/${TAG}_bench_"+sys.argv[1]+".err").read()[-2500:])
PY
}
bench p2 2; bench p1 1
ROMAN_HIP_LIBRARY=$PWD/roman_amd/csrc/variants/libW16.so bench w16p2 2
ROMAN_HIP_LIBRARY=$PWD/roman_amd/csrc/variants/libT.so timeout 100 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --latency-reps 2 --pipeline 1 > $OUT/${TAG}_timing.txt 2>$OUT/${TAG}_timing.err
grep -A4 "solve timing" $OUT/${TAG}_timing.err | tail -7
echo "== rocprofv3 kernel stats"
REPO=$PWD
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --cpu-sample 0 --latency-reps 5 --pipeline 1 > $OUT/${TAG}_prof_bench.txt 2>$OUT/${TAG}_prof.err ); echo "rocprof rc=$?"
F=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && python3 - "$F" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} max_us={float(r['MaxNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
find $OUT/${TAG}_prof -name "*kernel_trace.csv" -size +20M -delete
