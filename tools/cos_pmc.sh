#!/bin/bash
# MFMA pipe utilisation of the cosine kernels (separate PMC passes, kernel-trace only)
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; REPO=$PWD
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|SQ_BUSY_CY|GRBM_GUI|SQ_WAVES\b|SQ_INST_CYCLES_VMEM|TCP_TCC_READ_REQ_sum|TCC_HIT_sum|TCC_MISS_sum" | cut -c1-160 | sort -u | head -40 > $OUT/cos_pmc_list.txt
for mode in 0 16; do
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  g=$(echo $grp | tr ' ' '_' | cut -c1-30)
  export ROMAN_COS=$mode
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/cp_${mode}_$g -o b -- python $REPO/bench.py --steps 2 --warmup 1 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid > /dev/null 2>$OUT/cp_err.txt )
  F=$(find $OUT/cp_${mode}_$g -name "*counter_collection.csv" | head -1)
  python - "$F" "mode=$mode" <<'PY'
import csv, sys, collections
if not sys.argv[1]: print(sys.argv[2], "no counter file"); sys.exit()
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_cos' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items(): print(sys.argv[2], k, 'mean per launch', sum(v) / len(v), 'launches', len(v))
PY
  rm -rf $OUT/cp_${mode}_$g
done; done
tail -3 $OUT/cp_err.txt
