#!/bin/bash
# round-5 session J: k_lists (positions + kept-candidate lists from the upper blocks, one kernel) against the four kernels it stands for
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py tests/test_gpu_batch.py tests/test_gpu_shim.py tests/test_gpu_golden.py -q -x -m gpu -k "not fixed_point and not config4_full" > $OUT/r5j_pytest.txt 2>&1; echo "pytest rc=$?"; tail -8 $OUT/r5j_pytest.txt
for cfg in 0 1 0 1; do
  export ROMAN_LISTS=$cfg
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid --check-pairs 0 > /dev/null 2>&1 )
  F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
  python - "$F" "lists=$cfg" <<'PY'
import csv, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('k_lists', 'k_mirror', 'k_rowprefix', 'k_rowsort', 'k_upper', 'k_fill_list', 'k_solve_up<8')):
        print(sys.argv[2], n[:36].replace('void roman::', '').replace('roman::', ''), round(float(r['AverageNs']) / 1e3, 1), 'us', r['Calls'])
PY
  rm -rf $OUT/ab_tmp
  timeout 600 python bench.py --steps 40 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 256 --latency-reps 20 > $OUT/r5j_bench_$cfg.txt 2>$OUT/r5j_bench_$cfg.err
  echo "== lists=$cfg"; python tools/bench_digest.py $OUT/r5j_bench_$cfg.txt | head -1; python - <<PY
import json
d=json.loads(open("$OUT/r5j_bench_$cfg.txt").read().strip().splitlines()[-1]); print("   check", d["result_check"].get("oracle_identical"), "stage", {k: round(v,3) for k,v in d["roofline"]["isolated"]["stage_ms_per_call"].items()})
PY
done
