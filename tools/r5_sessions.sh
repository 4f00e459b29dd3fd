#!/bin/bash
# Round-5 GPU sessions, one per letter:  bash tools/r5_sessions.sh <a..t, z>   (repo root on an MI355X box; everything under gpurun_out/).
# The A/B sessions compare library builds kept under roman_amd/csrc/variants/ (git-ignored; rebuilt from the commits named in
# DESIGN.md 4.2 / profiles/r05/README.md): they are the record of what was measured, not something a fresh checkout can re-run as is.
S=$1
case "$S" in
a)
# round-5 session A: whole GPU suite (no fail-fast) on the new library, then old (libR4) vs new (libF1) on ONE box:
# per-kernel rocprofv3 averages and the bench line (20 steps, main measurement + latency only).
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 1500 python -m pytest tests -q -m gpu --durations=6 > $OUT/r5a_pytest.txt 2>&1; echo "pytest rc=$?"; tail -40 $OUT/r5a_pytest.txt
bash tools/ab_libs.sh roman_amd/csrc/variants/libR4.so roman_amd/csrc/variants/libF1.so k_solve_up k_count k_upper > $OUT/r5a_ab.txt 2>&1; cat $OUT/r5a_ab.txt
for L in libR4 libF1 libR4 libF1; do
  ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/$L.so timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 > $OUT/r5a_bench_$L.txt 2>$OUT/r5a_bench_$L.err
  echo "== $L"; python tools/bench_digest.py $OUT/r5a_bench_$L.txt | head -3
done
;;
b)
# round-5 session B: solver variants on ONE box — R4 (round 4), C2 (R4 + first ring loads in front of the barrier), F2 (fused passes
# + the same), each also with six quads in flight (ROMAN_SOLVE_DEEP=1): rocprofv3 average of k_solve_up (isolated launches) and the
# bench line (20 steps, three calls in flight; p50 of the single-pair call).
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py -q -m gpu -k "stagewise or config3 or demo_scale" > $OUT/r5b_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r5b_pytest.txt
for cfg in R4:0 C2:0 F2:0 C2:1 F2:1 R4:0 F2:0 C2:0; do
  L=${cfg%%:*}; DEEP=${cfg##*:}
  export ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/lib$L.so
  if [ "$DEEP" = "1" ]; then export ROMAN_SOLVE_DEEP=1; else unset ROMAN_SOLVE_DEEP; fi
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid --check-pairs 0 > /dev/null 2>&1 )
  F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
  python - "$F" "$L deep=$DEEP" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_solve_up<8' in r['Name']: print(sys.argv[2], r['Name'][:44], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
  rm -rf $OUT/ab_tmp
  timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps 20 > $OUT/r5b_bench_${L}_$DEEP.txt 2>$OUT/r5b_bench_${L}_$DEEP.err
  echo "== $L deep=$DEEP"; python tools/bench_digest.py $OUT/r5b_bench_${L}_$DEEP.txt | head -1
done
;;
c)
# round-5 session C: the restructured stream loop (F3: fused passes, split a compile-time constant, gathers one quad ahead, fixed LDS
# strides) against round 4 (R4) on ONE box + the solver-facing tests + the chunked host batch test.
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py tests/test_gpu_batch.py -q -m gpu -k "stagewise or config3 or demo_scale or fixed_point or large_host_batch or dense_matrix or ragged or tie_fallback or explicit_u0" > $OUT/r5c_pytest.txt 2>&1; echo "pytest rc=$?"; tail -15 $OUT/r5c_pytest.txt
for cfg in R4 F3 R4 F3; do
  L=$cfg
  export ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/lib$L.so
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid --check-pairs 0 > /dev/null 2>&1 )
  F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
  python - "$F" "$L" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_solve_up<8' in r['Name']: print(sys.argv[2], r['Name'][:44], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
  rm -rf $OUT/ab_tmp
  timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps 20 > $OUT/r5c_bench_${L}.txt 2>$OUT/r5c_bench_${L}.err
  echo "== $L"; python tools/bench_digest.py $OUT/r5c_bench_${L}.txt | head -3
done
;;
d)
# round-5 session D: per-phase cycle counters of the stream solver (timing build), B = 256 and B = 1
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/libT.so timeout 600 python bench.py --steps 2 --warmup 1 --pipeline 1 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps 2 > $OUT/r5d_bench.txt 2> $OUT/r5d_timing.txt
grep -A5 "solve timing" $OUT/r5d_timing.txt | tail -40
;;
e)
# round-5 session E: F3 (restructured stream) / F4 (+ division skip, next trial's first half in the objective's reduction) /
# F5 (+ element slots in use only) on ONE box: tests on the current tree, rocprofv3 average of k_solve_up, bench line, phase cycles.
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py tests/test_gpu_batch.py tests/test_u0_stability.py -q -m gpu -k "stagewise or config3 or config4_grid or demo_scale or fixed_point or dense_matrix or ragged or tie_fallback or explicit_u0 or random_start" > $OUT/r5e_pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 $OUT/r5e_pytest.txt
for L in F3 F4 F5 F3 F4 F5; do
  export ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/lib$L.so
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid --check-pairs 0 > /dev/null 2>&1 )
  F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
  python - "$F" "$L" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_solve_up<8' in r['Name']: print(sys.argv[2], r['Name'][:44], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
  rm -rf $OUT/ab_tmp
  timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps 20 > $OUT/r5e_bench_${L}.txt 2>$OUT/r5e_bench_${L}.err
  echo "== $L"; python tools/bench_digest.py $OUT/r5e_bench_${L}.txt | head -1
done
ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/libT.so timeout 600 python bench.py --steps 2 --warmup 1 --pipeline 1 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps 2 > $OUT/r5e_benchT.txt 2> $OUT/r5e_timing.txt
grep -A5 "solve timing" $OUT/r5e_timing.txt | grep -v "^--" | sed -n '1,6p;$p'
;;
f)
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
;;
g)
# round-5 session G: the whole GPU suite, smoke, the driver's bench command, rocprofv3 kernel stats + the four PMC groups
bash tools/gpu_session.sh r5g tests smoke bench prof pmc
;;
h)
# round-5 session H: k_count with the shared table gather (rows of a wave with the same map-1 object) on / off: parity subset, kernel averages, bench
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py -q -m gpu -k "stagewise or config3 or threshold" > $OUT/r5h_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r5h_pytest.txt
for cfg in 0 1 0 1; do
  export ROMAN_COUNT_SHARE=$cfg
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid --check-pairs 0 > /dev/null 2>&1 )
  F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
  python - "$F" "share=$cfg" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_count<1, 2, false' in r['Name'] or 'k_solve_up<8' in r['Name']: print(sys.argv[2], r['Name'][:40], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
  rm -rf $OUT/ab_tmp
  timeout 600 python bench.py --steps 40 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps 10 > $OUT/r5h_bench_$cfg.txt 2>$OUT/r5h_bench_$cfg.err
  echo "== share=$cfg"; python tools/bench_digest.py $OUT/r5h_bench_$cfg.txt | head -1
done
;;
i)
# round-5 session I: F6 (the committed solver) against F7 (+ the trial vector's sums reduced on the publish barrier: four workgroup
# barriers per pass instead of five), and F7 with four quads in flight (ROMAN_SOLVE_DEEP=1): tests, kernel averages, bench, phase cycles
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py tests/test_gpu_batch.py tests/test_u0_stability.py -q -m gpu -k "stagewise or config3 or config4_grid or demo_scale or fixed_point or dense_matrix or ragged or tie_fallback or explicit_u0 or random_start" > $OUT/r5i_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $OUT/r5i_pytest.txt
for cfg in F6:0 F7:0 F7:1 F6:0 F7:0 F7:1; do
  L=${cfg%%:*}; DEEP=${cfg##*:}
  export ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/lib$L.so
  if [ "$DEEP" = "1" ]; then export ROMAN_SOLVE_DEEP=1; else unset ROMAN_SOLVE_DEEP; fi
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid --check-pairs 0 > /dev/null 2>&1 )
  F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
  python - "$F" "$L deep=$DEEP" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_solve_up<8' in r['Name']: print(sys.argv[2], r['Name'][:44], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
  rm -rf $OUT/ab_tmp
  timeout 600 python bench.py --steps 40 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps 20 > $OUT/r5i_bench_${L}_$DEEP.txt 2>$OUT/r5i_bench_${L}_$DEEP.err
  echo "== $L deep=$DEEP"; python tools/bench_digest.py $OUT/r5i_bench_${L}_$DEEP.txt | head -1
done
unset ROMAN_SOLVE_DEEP
ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/libT.so timeout 600 python bench.py --steps 2 --warmup 1 --pipeline 1 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps 2 > $OUT/r5i_benchT.txt 2> $OUT/r5i_timing.txt
grep -A5 "solve timing" $OUT/r5i_timing.txt | grep -v "^--" | sed -n '1,6p;$p'
;;
j)
# round-5 session J: k_lists (positions + kept-candidate lists from the upper blocks, one kernel) against the four kernels it stands for
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py tests/test_gpu_batch.py tests/test_gpu_shim.py tests/test_gpu_golden.py -q -x -m gpu -k "not fixed_point and not config4_full" > $OUT/r5j_pytest.txt 2>&1; echo "pytest rc=$?"; tail -8 $OUT/r5j_pytest.txt
for cfg in 0 1 0 1; do
  export ROMAN_LISTS=$cfg
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid --check-pairs 0 > /dev/null 2>&1 )
  F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
  python - "$F" "lists=$cfg" <<'PY'
import csv, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('k_lists', 'k_mirror', 'k_rowprefix', 'k_rowsort', 'k_upper', 'k_fill_list', 'k_solve_up<8')):
        print(sys.argv[2], n[:36].replace('void roman::', '').replace('roman::', ''), round(float(r['AverageNs']) / 1e3, 1), 'us', r['Calls'])
PY
  rm -rf $OUT/ab_tmp
  timeout 600 python bench.py --steps 40 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 256 --latency-reps 20 > $OUT/r5j_bench_$cfg.txt 2>$OUT/r5j_bench_$cfg.err
  echo "== lists=$cfg"; python tools/bench_digest.py $OUT/r5j_bench_$cfg.txt | head -1; python - <<PY
import json
d=json.loads(open("$OUT/r5j_bench_$cfg.txt").read().strip().splitlines()[-1]); print("   check", d["result_check"].get("oracle_identical"), "stage", {k: round(v,3) for k,v in d["roofline"]["isolated"]["stage_ms_per_call"].items()})
PY
done
;;
k)
# round-5 session K: k_lists with the words of two steps in flight; ladder under both list paths; A/B and p50
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py tests/test_gpu_batch.py -q -x -m gpu -k "stagewise or config3 or config4_grid or demo_scale or mixed or large_host_batch or align_resident or ragged" > $OUT/r5k_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $OUT/r5k_pytest.txt
for cfg in 0 1 x; do
  if [ "$cfg" = "x" ]; then unset ROMAN_LISTS; else export ROMAN_LISTS=$cfg; fi
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid --check-pairs 0 > /dev/null 2>&1 )
  F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
  python - "$F" "lists=$cfg" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('k_lists', 'k_mirror', 'k_upper', 'k_count<1, 2, f')):
        print(sys.argv[2], n[:36].replace('void roman::', '').replace('roman::', ''), round(float(r['AverageNs']) / 1e3, 1), 'us', r['Calls'])
PY
  rm -rf $OUT/ab_tmp
  timeout 600 python bench.py --steps 40 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 256 --latency-reps 20 > $OUT/r5k_bench_$cfg.txt 2>$OUT/r5k_bench_$cfg.err
  echo "== lists=$cfg"; python tools/bench_digest.py $OUT/r5k_bench_$cfg.txt | head -1
done
;;
l)
# round-5 session L: k_lists, second cut (six steps in flight, one table read per column, two bits per round, wave-local sort stages)
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py tests/test_gpu_batch.py -q -x -m gpu -k "stagewise or config3 or config4_grid or demo_scale or mixed or large_host_batch or align_resident or ragged" > $OUT/r5l_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $OUT/r5l_pytest.txt
ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/libLT.so timeout 300 python bench.py --steps 1 --warmup 1 --pipeline 1 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps -1 2>&1 | grep "k_lists" | tail -4
for cfg in 0 x 0 x; do
  if [ "$cfg" = "x" ]; then unset ROMAN_LISTS; else export ROMAN_LISTS=$cfg; fi
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid --check-pairs 0 > /dev/null 2>&1 )
  F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
  python - "$F" "lists=$cfg" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('k_lists', 'k_upper')):
        print(sys.argv[2], n[:36].replace('void roman::', '').replace('roman::', ''), round(float(r['AverageNs']) / 1e3, 1), 'us', r['Calls'])
PY
  rm -rf $OUT/ab_tmp
  timeout 600 python bench.py --steps 40 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 256 --latency-reps 20 > $OUT/r5l_bench_$cfg.txt 2>$OUT/r5l_bench_$cfg.err
  echo "== lists=$cfg"; python tools/bench_digest.py $OUT/r5l_bench_$cfg.txt | head -1
done
;;
n)
# round-5 session N: the stream solver's waves take interleaved quads (default) against contiguous ranges (ROMAN_SOLVE_STRIDED=0),
# one build, one box: solver-facing tests under the new default, kernel averages, bench, phase cycles under both
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py tests/test_gpu_batch.py tests/test_u0_stability.py -q -m gpu -k "stagewise or config3 or config4_grid or demo_scale or fixed_point or dense_matrix or ragged or tie_fallback or explicit_u0 or random_start" > $OUT/r5n_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $OUT/r5n_pytest.txt
for ST in 0 1 0 1; do
  export ROMAN_SOLVE_STRIDED=$ST
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid --check-pairs 0 > /dev/null 2>&1 )
  F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
  python - "$F" "strided=$ST" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_solve_up<8' in r['Name']: print(sys.argv[2], r['Name'][:44], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
  rm -rf $OUT/ab_tmp
  timeout 600 python bench.py --steps 40 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 64 --latency-reps 20 > $OUT/r5n_bench_$ST.txt 2>$OUT/r5n_bench_$ST.err
  echo "== strided=$ST"; python tools/bench_digest.py $OUT/r5n_bench_$ST.txt | head -1
done
for ST in 0 1; do
  ROMAN_SOLVE_STRIDED=$ST ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/libT.so timeout 600 python bench.py --steps 2 --warmup 1 --pipeline 1 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps 2 > $OUT/r5n_benchT_$ST.txt 2> $OUT/r5n_timing_$ST.txt
  echo "== phases strided=$ST"; grep -A5 "solve timing" $OUT/r5n_timing_$ST.txt | grep -v "^--" | sed -n '1,6p;$p'
done
;;
o)
# round-5 session O/P: one demo-size pair per call (the serial caller at the reference's own scale).  Parts of the call's time
# (tools/gpu_demo_latency.py) with one wave per cosine block (k_cos_block, default for a few problems) and with one wave per
# problem (ROMAN_COS_BLOCK=0); phase cycles of k_small (variants/libST.so) for single pairs and inside the 4096-pair batch
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py -q -m gpu -k "cosine or demo_scale" > $OUT/r5p_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $OUT/r5p_pytest.txt
for CB in 0 1 0 1; do
  echo "== ROMAN_COS_BLOCK=$CB"; ROMAN_COS_BLOCK=$CB python tools/gpu_demo_latency.py 24 2>&1 | tee -a $OUT/r5p_demo_latency_$CB.txt
done
echo "== k_small phases, single pairs"
ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/libST.so python tools/gpu_demo_latency.py 4 2>&1 | grep "k_small" | sort | uniq -c | sort -rn | head -12
echo "== k_small phases, 4096 pairs per call"
ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/libST.so python tools/gpu_demo_scale.py 2>&1 | grep "k_small" | tail -16
;;
q)
# round-5 session Q: the solvers' shared tail (selection of the omega largest, cross-covariance, rotation) rebuilt — ranks from LDS
# only, the sums by one wave with DPP, the Jacobi rotation in registers: whole GPU suite, the tail's phase cycles (variants/libST.so),
# bench line, demo-scale latency and rate
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 1500 python -m pytest tests -q -m gpu -x --durations=5 > $OUT/r5q_pytest.txt 2>&1; echo "pytest rc=$?"; tail -12 $OUT/r5q_pytest.txt
( export ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/libST.so
  python tools/gpu_demo_latency.py 4 2>&1 | grep "finish_one\|k_small" | sort | uniq -c | sort -rn | head -4
  python tools/gpu_demo_scale.py 2>&1 | grep "finish_one\|k_small" | tail -6
  python bench.py --steps 2 --warmup 1 --pipeline 1 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps 2 2>&1 | grep "finish_one" | tail -4 )
timeout 900 python bench.py --steps 40 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 256 --latency-reps 20 > $OUT/r5q_bench.txt 2>$OUT/r5q_bench.err
python tools/bench_digest.py $OUT/r5q_bench.txt | head -4
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps -1 --cpu-sample 0 --no-extras --no-grid --check-pairs 0 > /dev/null 2>&1 )
F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_solve_up<8' in r['Name']: print(r['Name'][:44], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
rm -rf $OUT/ab_tmp
python tools/gpu_demo_latency.py 24 2>&1 | tee $OUT/r5q_demo_latency.txt
python tools/gpu_demo_scale.py 2>&1 | tail -12 | tee $OUT/r5q_demo_scale.txt
;;
r)
# round-5 session R: the tree with the rebuilt tail against the tree before it (variants/libR5m.so = commit 62e4741) on ONE box,
# alternating: bench line (40 steps) and the isolated launch of k_solve_up; phase cycles incl. set-up + tail (variants/libT.so);
# k_cos_block against k_cos_wave by batch size
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
for L in R5m new R5m new; do
  if [ "$L" = "new" ]; then unset ROMAN_HIP_LIBRARY; else export ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/lib$L.so; fi
  timeout 600 python bench.py --steps 40 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps 20 > $OUT/r5r_bench_$L.txt 2>$OUT/r5r_bench_$L.err
  echo "== $L"; python tools/bench_digest.py $OUT/r5r_bench_$L.txt | head -1
  python - $OUT/r5r_bench_$L.txt <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   isolated launch", d["roofline"]["isolated"]["avg_launch_ms"], "ms, roofline frac isolated", round(d["roofline"]["isolated"]["frac"], 3), "| B=1 stages", d["latency_breakdown"]["stage_ms"])
PY
done
unset ROMAN_HIP_LIBRARY
ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/libT.so timeout 600 python bench.py --steps 2 --warmup 1 --pipeline 1 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps 2 > $OUT/r5r_benchT.txt 2> $OUT/r5r_timing.txt
grep -A5 "solve timing" $OUT/r5r_timing.txt | grep -v "^--" | sed -n '1,6p;$p'
python tools/gpu_cos_block_sweep.py 2>&1 | tee $OUT/r5r_cos_block_sweep.txt
;;
s)
# round-5 session S: the tail once more — the ranks by groups of lanes, the compaction's counter once per wave, one division and
# one square root less per Jacobi step: whole GPU suite, the tail's phase cycles (variants/libST.so), and the tree of session Z
# (variants/libR5z.so) against this one on ONE box, alternating
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/r5s_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r5s_pytest.txt
( export ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/libST.so
  python tools/gpu_demo_latency.py 4 2>&1 | grep "finish_one" | sort | uniq -c | sort -rn | head -2
  python bench.py --steps 2 --warmup 1 --pipeline 1 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps 2 2>&1 | grep "finish_one" | tail -3 )
for L in R5z new R5z new; do
  if [ "$L" = "new" ]; then unset ROMAN_HIP_LIBRARY; else export ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/lib$L.so; fi
  timeout 600 python bench.py --steps 40 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 64 --latency-reps 20 > $OUT/r5s_bench_$L.txt 2>$OUT/r5s_bench_$L.err
  echo "== $L"; python tools/bench_digest.py $OUT/r5s_bench_$L.txt | head -1
  python - $OUT/r5s_bench_$L.txt <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   isolated launch", d["roofline"]["isolated"]["avg_launch_ms"], "ms | B=1 stages", d["latency_breakdown"]["stage_ms"], "| identical", d["result_check"].get("oracle_identical"))
PY
done
;;
t)
# round-5 session T: positions of the stream layout without a sort (place_keys: histogram of the degrees, scan, ranks inside the
# ranges of equal degree) in k_lists and k_rowsort: whole GPU suite, then the tree before it (variants/libR5s.so) against this one
# on ONE box, alternating: bench line, p50, and the kernels' rocprofv3 averages (k_lists in the batch, k_rowsort in the single-pair call)
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/r5t_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r5t_pytest.txt
for L in R5s new R5s new; do
  if [ "$L" = "new" ]; then unset ROMAN_HIP_LIBRARY; else export ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/lib$L.so; fi
  timeout 600 python bench.py --steps 40 --warmup 5 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps 20 > $OUT/r5t_bench_$L.txt 2>$OUT/r5t_bench_$L.err
  echo "== $L"; python tools/bench_digest.py $OUT/r5t_bench_$L.txt | head -1
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_tmp -o b -- python $REPO/bench.py --steps 5 --warmup 2 --pipeline 1 --latency-reps 100 --cpu-sample 0 --no-extras --no-grid --check-pairs 0 > /dev/null 2>&1 )
  F=$(find $OUT/ab_tmp -name "*kernel_stats.csv" | head -1)
  python - "$F" "$L" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_lists' in r['Name'] or 'k_rowsort' in r['Name']: print('  ', sys.argv[2], r['Name'][:24], r['Calls'], 'calls, avg', round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
  rm -rf $OUT/ab_tmp
done
;;
z)
# round-5 session Z (final tree, after the tail rewrite): the whole GPU suite, smoke, the driver's bench command, rocprofv3 kernel
# stats + the four PMC groups, solver phase cycles, the serial caller at demo scale, the N > 1 code path on one GPU
bash tools/gpu_session.sh r5z tests smoke bench prof pmc
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; REPO=$PWD
ROMAN_HIP_LIBRARY=$REPO/roman_amd/csrc/variants/libT.so timeout 600 python bench.py --steps 2 --warmup 1 --pipeline 1 --no-extras --no-grid --cpu-sample 0 --check-pairs 0 --latency-reps 2 > $OUT/r5z_benchT.txt 2> $OUT/r5z_timing.txt
grep -A5 "solve timing" $OUT/r5z_timing.txt | grep -v "^--" | sed -n '1,6p;$p' | tee $OUT/r5z_phase_cycles.txt
python tools/gpu_demo_latency.py 24 2>&1 | tee $OUT/r5z_demo_latency.txt
bash tools/scale_preflight.sh 2>&1 | tail -6
;;
m)
# round-5 session M (final tree): the whole GPU suite, smoke, the driver's bench command, rocprofv3 kernel stats + the four PMC groups, solver phase cycles
bash tools/gpu_session.sh r5m tests smoke bench prof pmc
;;
*) echo "usage: bash tools/r5_sessions.sh <a..m>"; exit 2 ;;
esac
