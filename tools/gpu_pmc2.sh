#!/bin/bash
# Focused PMC passes on the B=256 launches only (no B=1 latency probes).  usage: bash tools/gpu_pmc2.sh tag
TAG=${1:-pmc2}; OUT=$PWD/gpurun_out/${TAG}_pmc; REPO=$PWD; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- \
      python $REPO/bench.py --steps 2 --warmup 1 --cpu-sample 0 --latency-reps -1 --no-profile --pipeline 1 > $OUT/$name.log 2>&1 ); echo "$name rc=$?"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
run sq3 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU
run grbm GRBM_GUI_ACTIVE
python3 $REPO/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
find $OUT -name "*.csv" -size +8M -delete
