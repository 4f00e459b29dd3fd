#!/bin/bash
export TMPDIR=/tmp ROMAN_DEBUG=1
echo "== the hanging test, stage prints"; timeout 80 python -X faulthandler -m pytest tests/test_gpu_batch.py -x -q -s -m gpu -k large_live -o faulthandler_timeout=50 2>&1 | grep -v "^  File" | tail -45; echo "rc=$?"
