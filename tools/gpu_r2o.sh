#!/bin/bash
TAG=${1:-r2o}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== coop n=60 (L=3600)"; timeout 40 python -X faulthandler tools/gpu_large_case.py 60 2>&1 | tail -9; echo "rc=$?"
echo "== coop n=100"; timeout 40 python -X faulthandler tools/gpu_large_case.py 100 2>&1 | tail -9; echo "rc=$?"
echo "== coop n=200"; timeout 120 python -X faulthandler tools/gpu_large_case.py 200 2>&1 | tail -9; echo "rc=$?"
echo "== fallback tests"; timeout 200 python -X faulthandler -m pytest -o faulthandler_timeout=90 tests/test_gpu_batch.py tests/test_gpu_full_configs.py -q -x -m gpu -k "large_live or mixed or ragged or dense_matrix" 2>&1 | tail -8
