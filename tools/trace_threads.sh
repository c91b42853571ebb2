#!/bin/bash
# Kernel durations of single-object ops alone (he_bench_driver single) and under 1 / 4 / 16 / 64 host threads (he_bench_driver threads) with the mirror's stream
# set fixed at TROY_STREAMS (default 8): does concurrency stretch the kernels, or is it the dispatch path that saturates?  -> gpurun_out/<tag>_thr_{single,threads}_trace.txt
TAG=${1:-r06}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export TROY_STREAMS=${TROY_STREAMS:-8}
for MODE in single threads; do
  rocprofv3 --kernel-trace --stats -d "$OUT/prof_thr/trace" -o bench -- $ROOT/tests/cpp/he_bench_driver $MODE > "$OUT/${TAG}_thr_$MODE.log" 2>&1
  python3 $ROOT/tools/rocpd_summary.py "$OUT/prof_thr/trace/bench_results.db" > "$OUT/${TAG}_thr_${MODE}_trace.txt"
  rm -rf "$OUT/prof_thr"
done
