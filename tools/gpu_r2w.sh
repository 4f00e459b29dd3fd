#!/bin/bash
export TMPDIR=/tmp
timeout 40 python -X faulthandler tools/gpu_large_case.py 100 2>&1 | tail -6
timeout 60 python -X faulthandler tools/gpu_large_case.py 200 nocheck 2>&1 | tail -5
