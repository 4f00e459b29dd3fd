#!/bin/bash
# round-5 session D: per-phase cycle counters of the stream solver (timing build), B = 256 and B = 1
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/libT.so timeout 600 python bench.py --steps 2 --warmup 1 --pipeline 1 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps 2 > $OUT/r5d_bench.txt 2> $OUT/r5d_timing.txt
grep -A5 "solve timing" $OUT/r5d_timing.txt | tail -40
