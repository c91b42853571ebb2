#!/bin/bash
# Kernel trace of relinearize on the reference bench tool's default chain (N = 8192 {60,40,40,60}) -> gpurun_out/<tag>_mixed_trace.txt
set -e
TAG=${1:-mixed}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/prof_$TAG/trace" -o mixed -- python3 $ROOT/tools/profile_mixed.py > "$OUT/${TAG}_mixed.log" 2>&1
cd "$ROOT"
python3 tools/rocpd_summary.py "$OUT/prof_$TAG/trace/mixed_results.db" > "$OUT/${TAG}_mixed_trace.txt"
rm -rf "$OUT/prof_$TAG"
