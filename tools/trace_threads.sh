#!/bin/bash
ROOT=/root/repo
OUT="$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
export TROY_STREAMS=8
rocprofv3 --kernel-trace --stats -d "$OUT/prof_thr/trace" -o bench -- $ROOT/tests/cpp/he_bench_driver single > "$OUT/r06_thr_single.log" 2>&1
python3 $ROOT/tools/rocpd_summary.py "$OUT/prof_thr/trace/bench_results.db" > "$OUT/r06_thr_single_trace.txt"
rm -rf "$OUT/prof_thr"
rocprofv3 --kernel-trace --stats -d "$OUT/prof_thr/trace" -o bench -- $ROOT/tests/cpp/he_bench_driver threads > "$OUT/r06_thr_threads.log" 2>&1
python3 $ROOT/tools/rocpd_summary.py "$OUT/prof_thr/trace/bench_results.db" > "$OUT/r06_thr_threads_trace.txt"
rm -rf "$OUT/prof_thr"
