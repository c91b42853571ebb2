#!/bin/bash
# Kernel trace only of the default bench.py workload (seconds): tools/trace_bench.sh <tag> [bench args] -> gpurun_out/<tag>_trace.txt
set -e
TAG=${1:-trace}; shift || true
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/prof_$TAG/trace" -o bench -- python3 $ROOT/bench.py --steps 10 --warmup 1 --inner 4 --no-cpu-baseline --no-extra "$@" > "$OUT/${TAG}_bench.log" 2>&1
cd "$ROOT"
python3 tools/rocpd_summary.py "$OUT/prof_$TAG/trace/bench_results.db" > "$OUT/${TAG}_trace.txt"
rm -rf "$OUT/prof_$TAG"
