#!/bin/bash
# round-5 session B: solver variants on ONE box — R4 (round 4), C2 (R4 + first ring loads in front of the barrier), F2 (fused passes
# + the same), each also with six quads in flight (ROMAN_SOLVE_DEEP=1): rocprofv3 average of k_solve_up (isolated launches) and the
# bench line (20 steps, three calls in flight; p50 of the single-pair call).
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py -q -m gpu -k "stagewise or config3 or demo_scale" > $OUT/r5b_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r5b_pytest.txt
for cfg in R4:0 C2:0 F2:0 C2:1 F2:1 R4:0 F2:0 C2:0; do
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
  timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps 20 > $OUT/r5b_bench_${L}_$DEEP.txt 2>$OUT/r5b_bench_${L}_$DEEP.err
  echo "== $L deep=$DEEP"; python tools/bench_digest.py $OUT/r5b_bench_${L}_$DEEP.txt | head -1
done
