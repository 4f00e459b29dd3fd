#!/bin/bash
# round-2 first GPU session: parity suite, MFMA order probe, baseline bench
TAG=${1:-r2a}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
echo "== mfma order probe"; timeout 600 python tools/gpu_mfma_order.py > $OUT/${TAG}_mfma.txt 2>&1; echo "rc=$?"; tail -30 $OUT/${TAG}_mfma.txt
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -q -m gpu -x --durations=15 > $OUT/${TAG}_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -40 $OUT/${TAG}_pytest_gpu.txt
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.txt 2>$OUT/${TAG}_bench.err; echo "bench rc=$?"; tail -1 $OUT/${TAG}_bench.txt | cut -c1-1500; tail -3 $OUT/${TAG}_bench.err
