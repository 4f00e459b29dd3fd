#!/bin/bash
# round 6, second half: the exact gate of the prefiltered pair-test sweep with its operands from coordinates in LDS (ROMAN_COUNT_OBJ=1) against
# the tables and heights from memory (the default), alternating on ONE box: parity tests of the sweeps under both, then rocprofv3
# kernel averages and the bench line.   usage: bash tools/r6_kcount_obj_ab.sh <tag>
TAG=${1:-r6ko}
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
for o in 1 0; do
  export ROMAN_COUNT_OBJ=$o
  timeout 900 python -m pytest tests -q -x -m gpu -k "candidate or stagewise or threshold or far_from or config3 or golden or ladder" > $OUT/${TAG}_pytest_obj$o.txt 2>&1; echo "obj=$o pytest rc=$?"; tail -2 $OUT/${TAG}_pytest_obj$o.txt
done
for o in 1 0 1 0; do
  export ROMAN_COUNT_OBJ=$o
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid --check-pairs 0 > /dev/null 2>&1 )
  F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
  python - "$F" "obj=$o" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if ('k_count<' in r['Name'] or 'k_lists' in r['Name']) and int(r['Calls']) > 4: print(sys.argv[2], r['Name'][:40], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
  rm -rf $OUT/ab_tmp
  timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-grid --cpu-sample 0 --latency-reps 20 2>/dev/null | python tools/bench_digest.py /dev/stdin | head -2
done
# measured (profiles/r06/README.md): k_count 397 / 373 us (obj 1 / 0), twice; headline 162.9-163.1 / 163.7-164.3 k alignments/s.  Not kept as the default.
