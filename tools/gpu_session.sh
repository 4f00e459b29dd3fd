#!/bin/bash
# One GPU-box session, parameterised: runs the named steps in order, everything under gpurun_out/<tag>_*.
#   usage (repo root on the box): bash tools/gpu_session.sh <tag> step [step ...]
# steps:  tests[:<pytest -k expr>]   pytest -m gpu (-x), optional -k expression
#         smoke                      __graft_entry__.smoke()
#         bench[:<extra args>]       python bench.py <args>  (default: --steps 20 --warmup 5)
#         prof[:<bench args>]        rocprofv3 --kernel-trace --stats of bench.py <args> (default: B=256 launches only)
#         pmc[:<bench args>]         rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / two SQ sets incl. MFMA busy and LDS bank conflicts) of the
#                                    same (or of bench.py <args>), each pass on its own with --kernel-trace only, summarised by tools/pmc_summary.py
#         large[:<n>]                tools/gpu_large_case.py <n> (large-live-set solver, default 200)
#         ubench:<name>              tools/ubench/<name> (prebuilt binary travels with the snapshot)
#         py:<script and args>       python <script and args>
#         env:NAME=VALUE             export NAME=VALUE for the steps that follow (env:NAME= clears it)
TAG=$1; shift
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
for step in "$@"; do
  name=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  echo "=== [$TAG] $name $arg"
  case $name in
    tests) timeout 2400 python -m pytest tests -q -x -m gpu --durations=8 ${arg:+-k "$arg"} > $OUT/${TAG}_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -25 $OUT/${TAG}_pytest_gpu.txt ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $OUT/${TAG}_smoke.txt ;;
    bench) timeout 900 python bench.py ${arg:---steps 20 --warmup 5} > $OUT/${TAG}_bench.txt 2>$OUT/${TAG}_bench.err; echo "bench rc=$?"; tail -3 $OUT/${TAG}_bench.err
           cp bench_extras.json $OUT/${TAG}_bench_extras.json 2>/dev/null     # (the full record of THIS run: later steps' bench runs overwrite bench_extras.json)
           python tools/bench_digest.py $OUT/${TAG}_bench.txt ;;
    prof)  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $REPO/bench.py ${arg:---steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras} > $OUT/${TAG}_prof_bench.txt 2>$OUT/${TAG}_prof.err ); echo "rocprof rc=$?"
           F=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $OUT/${TAG}_kernel_stats.csv && head -24 "$F" | cut -c1-200
           find $OUT/${TAG}_prof -name "*kernel_trace.csv" -size +20M -delete ;;
    pmc)   for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU"; do
             g=$(echo $grp | tr ' ' '_' | cut -c1-24)
             ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/${TAG}_pmc/$g -o bench -- python $REPO/bench.py ${arg:---steps 3 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras} > /dev/null 2>$OUT/${TAG}_pmc_$g.err ); echo "pmc $g rc=$?"
             find $OUT/${TAG}_pmc/$g -name "*kernel_trace.csv" -size +20M -delete
           done
           python tools/pmc_summary.py $OUT/${TAG}_pmc > $OUT/${TAG}_pmc_summary.txt 2>&1; tail -40 $OUT/${TAG}_pmc_summary.txt ;;
    large) timeout 900 python tools/gpu_large_case.py ${arg:-200} > $OUT/${TAG}_large_${arg:-200}.txt 2>&1; echo "large rc=$?"; tail -12 $OUT/${TAG}_large_${arg:-200}.txt ;;
    ubench) timeout 300 tools/ubench/$arg > $OUT/${TAG}_ubench_$arg.txt 2>&1; echo "ubench rc=$?"; cat $OUT/${TAG}_ubench_$arg.txt ;;
    py)    timeout 1200 python $arg > $OUT/${TAG}_py_$(echo $arg | tr ' /' '__' | cut -c1-40).txt 2>&1; echo "py rc=$?"; tail -30 $OUT/${TAG}_py_$(echo $arg | tr ' /' '__' | cut -c1-40).txt ;;
    env)   export "$arg"; echo "exported $arg" ;;
    *) echo "unknown step $name" ;;
  esac
done
