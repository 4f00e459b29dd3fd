#!/bin/bash
# A/B of environment settings on ONE box: bench.py throughput, alternating.   usage: bash tools/ab_env.sh "<VAR=val ...>" "<VAR=val ...>" [reps]
REPO=$PWD
for rep in $(seq 1 ${3:-2}); do for E in "$1" "$2"; do
  echo -n "[$E] "
  env $E python $REPO/bench.py --steps 40 --warmup 10 --no-extras --no-grid --cpu-sample 0 --latency-reps -1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), 'alignments/s', d['ms_per_step'], 'ms/step')"
done; done
