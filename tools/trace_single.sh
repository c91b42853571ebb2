#!/bin/bash
# Kernel trace of single-object multiply + relinearize + rescale through troy::Evaluator (tests/cpp/he_bench_driver bench) -> gpurun_out/<tag>_single_trace.txt
set -e
TAG=${1:-single}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/prof_$TAG/trace" -o bench -- $ROOT/tests/cpp/he_bench_driver single > "$OUT/${TAG}_single.log" 2>&1
cd "$ROOT"
python3 tools/rocpd_summary.py "$OUT/prof_$TAG/trace/bench_results.db" > "$OUT/${TAG}_single_trace.txt"
rm -rf "$OUT/prof_$TAG"
