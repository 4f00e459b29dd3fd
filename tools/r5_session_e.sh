#!/bin/bash
# round-5 session E: F3 (restructured stream) / F4 (+ division skip, next trial's first half in the objective's reduction) /
# F5 (+ element slots in use only) on ONE box: tests on the current tree, rocprofv3 average of k_solve_up, bench line, phase cycles.
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py tests/test_gpu_batch.py tests/test_u0_stability.py -q -m gpu -k "stagewise or config3 or config4_grid or demo_scale or fixed_point or dense_matrix or ragged or tie_fallback or explicit_u0 or random_start" > $OUT/r5e_pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 $OUT/r5e_pytest.txt
for L in F3 F4 F5 F3 F4 F5; do
  export ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/lib$L.so
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid --check-pairs 0 > /dev/null 2>&1 )
  F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
  python - "$F" "$L" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_solve_up<8' in r['Name']: print(sys.argv[2], r['Name'][:44], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
  rm -rf $OUT/ab_tmp
  timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps 20 > $OUT/r5e_bench_${L}.txt 2>$OUT/r5e_bench_${L}.err
  echo "== $L"; python tools/bench_digest.py $OUT/r5e_bench_${L}.txt | head -1
done
ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/libT.so timeout 600 python bench.py --steps 2 --warmup 1 --pipeline 1 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps 2 > $OUT/r5e_benchT.txt 2> $OUT/r5e_timing.txt
grep -A5 "solve timing" $OUT/r5e_timing.txt | grep -v "^--" | sed -n '1,6p;$p'
