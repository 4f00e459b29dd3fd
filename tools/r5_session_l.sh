#!/bin/bash
# round-5 session L: k_lists, second cut (six steps in flight, one table read per column, two bits per round, wave-local sort stages)
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py tests/test_gpu_batch.py -q -x -m gpu -k "stagewise or config3 or config4_grid or demo_scale or mixed or large_host_batch or align_resident or ragged" > $OUT/r5l_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $OUT/r5l_pytest.txt
ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/libLT.so timeout 300 python bench.py --steps 1 --warmup 1 --pipeline 1 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps -1 2>&1 | grep "k_lists" | tail -4
for cfg in 0 x 0 x; do
  if [ "$cfg" = "x" ]; then unset ROMAN_LISTS; else export ROMAN_LISTS=$cfg; fi
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid --check-pairs 0 > /dev/null 2>&1 )
  F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
  python - "$F" "lists=$cfg" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('k_lists', 'k_upper')):
        print(sys.argv[2], n[:36].replace('void roman::', '').replace('roman::', ''), round(float(r['AverageNs']) / 1e3, 1), 'us', r['Calls'])
PY
  rm -rf $OUT/ab_tmp
  timeout 600 python bench.py --steps 40 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 256 --latency-reps 20 > $OUT/r5l_bench_$cfg.txt 2>$OUT/r5l_bench_$cfg.err
  echo "== lists=$cfg"; python tools/bench_digest.py $OUT/r5l_bench_$cfg.txt | head -1
done
