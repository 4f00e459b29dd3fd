#!/bin/bash
# round-5 session I: F6 (the committed solver) against F7 (+ the trial vector's sums reduced on the publish barrier: four workgroup
# barriers per pass instead of five), and F7 with four quads in flight (ROMAN_SOLVE_DEEP=1): tests, kernel averages, bench, phase cycles
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py tests/test_gpu_batch.py tests/test_u0_stability.py -q -m gpu -k "stagewise or config3 or config4_grid or demo_scale or fixed_point or dense_matrix or ragged or tie_fallback or explicit_u0 or random_start" > $OUT/r5i_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $OUT/r5i_pytest.txt
for cfg in F6:0 F7:0 F7:1 F6:0 F7:0 F7:1; do
  L=${cfg%%:*}; DEEP=${cfg##*:}
  export ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/lib$L.so
  if [ "$DEEP" = "1" ]; then export ROMAN_SOLVE_DEEP=1; else unset ROMAN_SOLVE_DEEP; fi
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid --check-pairs 0 > /dev/null 2>&1 )
  F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
  python - "$F" "$L deep=$DEEP" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_solve_up<8' in r['Name']: print(sys.argv[2], r['Name'][:44], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
  rm -rf $OUT/ab_tmp
  timeout 600 python bench.py --steps 40 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps 20 > $OUT/r5i_bench_${L}_$DEEP.txt 2>$OUT/r5i_bench_${L}_$DEEP.err
  echo "== $L deep=$DEEP"; python tools/bench_digest.py $OUT/r5i_bench_${L}_$DEEP.txt | head -1
done
unset ROMAN_SOLVE_DEEP
ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/libT.so timeout 600 python bench.py --steps 2 --warmup 1 --pipeline 1 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps 2 > $OUT/r5i_benchT.txt 2> $OUT/r5i_timing.txt
grep -A5 "solve timing" $OUT/r5i_timing.txt | grep -v "^--" | sed -n '1,6p;$p'
