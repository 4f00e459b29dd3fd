#!/bin/bash
# Kernel trace of the pipelined bench + overlap analysis.  usage: bash tools/gpu_trace.sh tag [bench args]
TAG=${1:-t}; shift; OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT; export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_trace -o t -- python $REPO/bench.py --steps 12 --warmup 2 --cpu-sample 0 --latency-reps -1 --no-profile "$@" > $OUT/${TAG}_trace.txt 2>$OUT/${TAG}_trace.err )
F=$(find $OUT/${TAG}_trace -name "*kernel_trace.csv" | head -1)
python3 $REPO/tools/trace_overlap.py $F
tail -1 $OUT/${TAG}_trace.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', round(d['value']), 'ms/step', round(d['ms_per_step'],3))"
rm -rf $OUT/${TAG}_trace
