#!/bin/bash
# PMC passes over tools/ubench/cos_sel_time (k_cos_sel alone): SEL_ONLY picks the candidate densities.
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
for only in 0 0.05; do
  export SEL_ONLY=$only
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_IFETCH"; do
    i=$((i+1))
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/cs_pmc_${only}_$i -o p -- $REPO/tools/ubench/cos_sel_time > /dev/null 2>$OUT/cs_pmc_${only}_$i.err ); echo "pmc $only/$i rc=$?"
  done
done
python - <<'P'
import csv, glob, collections, os
for only in ("0", "0.05"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/cs_pmc_{only}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_cos_sel" in k: agg["k_cos_sel"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== SEL_ONLY", only)
    for c, v in sorted(agg["k_cos_sel"].items()):
        print(f"   {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
P
