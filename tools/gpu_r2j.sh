#!/bin/bash
TAG=${1:-r2j}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$PWD
echo "== sharded device path test"; timeout 200 python -m pytest tests/test_gpu_batch.py -q -x -m gpu -k "sharded" 2>&1 | tail -5
echo "== grid workload, 1 GPU"; timeout 240 python bench.py --workload grid --steps 2 --warmup 1 --cpu-sample 0 --latency-reps 3 > $OUT/${TAG}_grid.txt 2>$OUT/${TAG}_grid.err; echo "rc=$?"
python - <<PY
import json
try:
    d=json.loads(open("$OUT/${TAG}_grid.txt").read().strip().splitlines()[-1])
    print("grid value", round(d["value"]), "ms/step", round(d["ms_per_step"],2), d["config"]["workload"][:90], d["result_check"])
except Exception as e:
    print("grid bench failed", e); print(open("$OUT/${TAG}_grid.err").read()[-2500:])
PY
echo "== fill timing"; ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/libF.so timeout 100 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --latency-reps 1 --pipeline 1 > $OUT/${TAG}_ftiming.txt 2>$OUT/${TAG}_ftiming.err; grep "fill timing" $OUT/${TAG}_ftiming.err | tail -3
pmc() {  # name, counters...
  local name=$1; shift
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/${TAG}_pmc/$name -o p -- \
      python $REPO/bench.py --steps 2 --warmup 1 --cpu-sample 0 --latency-reps -1 --no-profile --pipeline 1 > $OUT/${TAG}_pmc_$name.log 2>&1 ); echo "pmc $name rc=$?"
}
pmc tcc1 FETCH_SIZE
pmc tcc2 WRITE_SIZE
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
python3 tools/pmc_summary.py $OUT/${TAG}_pmc > $OUT/${TAG}_pmc_summary.txt 2>&1; head -60 $OUT/${TAG}_pmc_summary.txt
find $OUT/${TAG}_pmc -name "*.csv" -size +8M -delete
echo "== default bench (driver flags)"; timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver.txt 2>$OUT/${TAG}_bench_driver.err; echo "rc=$?"; tail -c 3000 $OUT/${TAG}_bench_driver.txt
