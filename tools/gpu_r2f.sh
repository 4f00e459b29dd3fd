#!/bin/bash
export TMPDIR=/tmp ROMAN_DEBUG=1 ROMAN_HIP_LIBRARY=$PWD/roman_amd/csrc/variants/libD.so
echo "== small stream problem, forced overflow"; ROMAN_TEST_CAPNNZ=256 timeout 40 python -u tools/gpu_overflow_probe.py small 2>&1 | tail -40; echo "rc=$?"
echo "== large fallback problem, natural overflow"; timeout 60 python -u tools/gpu_overflow_probe.py large 2>&1 | tail -40; echo "rc=$?"
