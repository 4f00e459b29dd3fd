#!/bin/bash
# round-5 session A: whole GPU suite (no fail-fast) on the new library, then old (libR4) vs new (libF1) on ONE box:
# per-kernel rocprofv3 averages and the bench line (20 steps, main measurement + latency only).
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 1500 python -m pytest tests -q -m gpu --durations=6 > $OUT/r5a_pytest.txt 2>&1; echo "pytest rc=$?"; tail -40 $OUT/r5a_pytest.txt
bash tools/ab_libs.sh roman_amd/csrc/variants/libR4.so roman_amd/csrc/variants/libF1.so k_solve_up k_count k_upper > $OUT/r5a_ab.txt 2>&1; cat $OUT/r5a_ab.txt
for L in libR4 libF1 libR4 libF1; do
  ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/$L.so timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 > $OUT/r5a_bench_$L.txt 2>$OUT/r5a_bench_$L.err
  echo "== $L"; python tools/bench_digest.py $OUT/r5a_bench_$L.txt | head -3
done
