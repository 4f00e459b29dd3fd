#!/bin/bash
# Does a process that loaded the library die at exit under rocprofv3 --kernel-trace?  (round 6: the bench's own PMC passes did, after the
# whole-device solver got a third instantiation)  usage (GPU box, repo root): bash tools/r6_rocprof_exit.sh
export TMPDIR=/tmp
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
run() {   # tag, library ("" = the in-tree one), command...
  local tag=$1 lib=$2; shift 2
  ( cd /tmp && ROMAN_HIP_LIBRARY=$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpx_$tag -o t -- "$@" > $OUT/rpx_$tag.out 2>&1; echo "$tag: rc=$? segv=$(grep -c SIGSEGV $OUT/rpx_$tag.out) kernels=$(cat /tmp/rpx_$tag/*kernel_stats.csv 2>/dev/null | wc -l)" )
}
PY="import sys; sys.path.insert(0, '$REPO');"
run smoke "" python -c "$PY import __graft_entry__ as g; g.smoke()"
run smoke_old "$REPO/roman_amd/csrc/variants/libOLD.so" python -c "$PY import __graft_entry__ as g; g.smoke()"
run bench "" python $REPO/bench.py --steps 3 --warmup 1 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras
run bench_old "$REPO/roman_amd/csrc/variants/libOLD.so" python $REPO/bench.py --steps 3 --warmup 1 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras
run bench_p3 "" python $REPO/bench.py --steps 3 --warmup 1 --latency-reps -1 --cpu-sample 0 --no-extras
