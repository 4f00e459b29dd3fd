#!/bin/bash
# round-6 A/B of the pair-test kernel on ONE box: parity tests of both sweeps, then rocprofv3 kernel averages and the bench line for
#   PRE=1 (candidate generation, the default; 12 waves per workgroup), P16 (the same with 16 waves: variants/libP16.so), PRE=0 (plain sweep),
# alternating.   usage: bash tools/r6_kcount_ab.sh <tag> [configs...]
TAG=${1:-r6k}; shift; CFGS=${@:-"1 P16 0 1 P16 0"}
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 900 python -m pytest tests -q -x -m gpu -k "candidate or stagewise or threshold or far_from or config3" > $OUT/${TAG}_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $OUT/${TAG}_pytest.txt
for p in $CFGS; do
  unset ROMAN_HIP_LIBRARY; export ROMAN_COUNT_PRE=1
  case $p in
    0) export ROMAN_COUNT_PRE=0 ;;
    1) ;;
    *) export ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/lib$p.so ;;
  esac
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid --check-pairs 0 > /dev/null 2>&1 )
  F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
  python - "$F" "cfg=$p" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_count<' in r['Name'] and int(r['Calls']) > 4: print(sys.argv[2], r['Name'][:40], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
  cp "$F" $OUT/${TAG}_kernel_stats_cfg$p.csv
  rm -rf $OUT/ab_tmp
  timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps 20 2>/dev/null | python tools/bench_digest.py /dev/stdin | head -1
done
