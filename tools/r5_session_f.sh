#!/bin/bash
# round-5 session F: the package's pipelined entries on the GPU (tests) + the full bench line with the caller legs
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_submap_align.py -q -m gpu -k "align_resident or large_host_batch or sharded or submap" > $OUT/r5f_pytest.txt 2>&1; echo "pytest rc=$?"; tail -25 $OUT/r5f_pytest.txt
t0=$(date +%s)
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/r5f_bench.txt 2>$OUT/r5f_bench.err; echo "bench rc=$? in $(( $(date +%s) - t0 )) s"; tail -5 $OUT/r5f_bench.err
python tools/bench_digest.py $OUT/r5f_bench.txt | head -4
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5f_bench.txt").read().strip().splitlines()[-1])
print("caller_legs_error", d.get("caller_legs_error"), "side_legs_error", d.get("side_legs_error"))
print(json.dumps(d.get("caller"), indent=1)[:6000])
PY
