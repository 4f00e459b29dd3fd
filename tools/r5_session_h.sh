#!/bin/bash
# round-5 session H: k_count with the shared table gather (rows of a wave with the same map-1 object) on / off: parity subset, kernel averages, bench
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py -q -m gpu -k "stagewise or config3 or threshold" > $OUT/r5h_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r5h_pytest.txt
for cfg in 0 1 0 1; do
  export ROMAN_COUNT_SHARE=$cfg
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid --check-pairs 0 > /dev/null 2>&1 )
  F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
  python - "$F" "share=$cfg" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_count<1, 2, false' in r['Name'] or 'k_solve_up<8' in r['Name']: print(sys.argv[2], r['Name'][:40], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
  rm -rf $OUT/ab_tmp
  timeout 600 python bench.py --steps 40 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps 10 > $OUT/r5h_bench_$cfg.txt 2>$OUT/r5h_bench_$cfg.err
  echo "== share=$cfg"; python tools/bench_digest.py $OUT/r5h_bench_$cfg.txt | head -1
done
