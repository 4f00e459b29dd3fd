#!/bin/bash
TAG=${1:-r2b}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
echo "== lds ubench"; timeout 300 tools/ubench/lds_scatter > $OUT/${TAG}_lds.txt 2>&1; echo "rc=$?"; cat $OUT/${TAG}_lds.txt
echo "== diag cfg3"; timeout 600 python tools/gpu_diag_cfg3.py > $OUT/${TAG}_diag.txt 2>&1; echo "rc=$?"; cat $OUT/${TAG}_diag.txt
echo "== pytest -m gpu (all)"; timeout 2400 python -m pytest tests -q -m gpu --durations=8 > $OUT/${TAG}_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -60 $OUT/${TAG}_pytest_gpu.txt
