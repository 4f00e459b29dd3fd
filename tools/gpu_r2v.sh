#!/bin/bash
# rocprofv3 PMC passes on the final tree (one counter group per pass; no trace domains besides --kernel-trace)
TAG=${1:-r2v}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$PWD
pmc() {  # name, counters...
  local name=$1; shift
  ( cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/${TAG}_pmc/$name -o p -- \
      python $REPO/bench.py --steps 2 --warmup 1 --cpu-sample 0 --latency-reps -1 --no-profile --pipeline 1 > $OUT/${TAG}_pmc_$name.log 2>&1 ); echo "pmc $name rc=$?"
}
pmc tcc1 FETCH_SIZE
pmc tcc2 WRITE_SIZE
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
python3 tools/pmc_summary.py $OUT/${TAG}_pmc > $OUT/${TAG}_pmc_summary.txt 2>&1; grep -A14 "k_solve_up\|k_fill_list\|== k_upper" $OUT/${TAG}_pmc_summary.txt | head -60
find $OUT/${TAG}_pmc -name "*.csv" -size +8M -delete
