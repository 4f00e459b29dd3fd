#!/bin/bash
# PMC counter collection for the bench workload (separate passes per counter group, as the
# MI355X guide prescribes).  usage: bash tools/gpu_pmc.sh [tag] [extra bench args]
TAG=${1:-pmc}; shift
OUT=$PWD/gpurun_out/${TAG}_pmc
REPO=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- \
      python $REPO/bench.py --steps 2 --warmup 1 --cpu-sample 0 --latency-reps 2 --no-profile "${EXTRA[@]}" > $OUT/$name.log 2>&1 )
  echo "$name rc=$?"
}
EXTRA=("$@")
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
run tcc3 TCC_HIT_sum TCC_MISS_sum
run grbm GRBM_GUI_ACTIVE
python3 - <<'PY'
import csv, glob, os, collections, sys
out=os.environ.get('OUT') or sys.argv[0]
PY
python3 $REPO/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
find $OUT -name "*.csv" -size +8M -delete
