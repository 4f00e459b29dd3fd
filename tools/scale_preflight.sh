#!/bin/bash
# Multi-GPU pre-flight on ONE GPU: bench.py through torch.distributed.run with --nproc-per-node 1 and the nccl (= RCCL)
# backend forced on, so that process-group init, the library's stream / torch's stream ordering, ctx.join() and the
# all_gather path of the N > 1 code run at least once before the driver's scaling sweep.  Also align_sharded's device path.
#   usage (repo root on a GPU box): bash tools/scale_preflight.sh [tag]
TAG=${1:-preflight}
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROMAN_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 \
    bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --cpu-sample 0 > $OUT/${TAG}_dist1.txt 2> $OUT/${TAG}_dist1.err
echo "torchrun bench (pairs) rc=$?"; python tools/bench_digest.py $OUT/${TAG}_dist1.txt
# the N > 1 headline workload (config 4, strong scaling: the grid dealt by deal_by_cost, ONE all_gather of byte records per call) at world size 1
ROMAN_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 \
    bench.py --gpus 1 --workload grid --steps 6 --warmup 2 --no-extras --cpu-sample 0 > $OUT/${TAG}_dist1_grid.txt 2> $OUT/${TAG}_dist1_grid.err
echo "torchrun bench (grid) rc=$?"; tail -3 $OUT/${TAG}_dist1_grid.err; cat $OUT/${TAG}_dist1_grid.txt | tail -1 | cut -c1-1500
timeout 300 python -m pytest tests -q -m gpu -k "align_sharded" 2>&1 | tail -3
