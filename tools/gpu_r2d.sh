#!/bin/bash
TAG=${1:-r2d}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== step debug big fallback problem"; timeout 150 python -u tools/gpu_step_debug.py clipper 150 150 0 61 2>&1 | tail -14; echo "rc=$?"
echo "== step debug medium fallback (gravity 70x70, L=4900)"; timeout 100 python -u tools/gpu_step_debug.py gravity 70 70 0 11 2>&1 | tail -14; echo "rc=$?"
echo "== the hanging test with a traceback dump"; timeout 120 python -X faulthandler -m pytest tests/test_gpu_batch.py -x -q -m gpu -k large_live -o faulthandler_timeout=40 2>&1 | tail -40; echo "rc=$?"
