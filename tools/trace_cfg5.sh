#!/bin/bash
# Kernel trace of BASELINE config 5 (512^3 packed BFV matmul through MatmulHelper, tests/cpp/matmul_driver) -> gpurun_out/<tag>_cfg5_trace.txt
set -e
TAG=${1:-cfg5}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/prof_$TAG/trace" -o m -- $ROOT/tests/cpp/matmul_driver 512 512 512 5 1 1 > "$OUT/${TAG}_cfg5.log" 2>&1
cd "$ROOT"
python3 tools/rocpd_summary.py "$OUT/prof_$TAG/trace/m_results.db" > "$OUT/${TAG}_cfg5_trace.txt"
rm -rf "$OUT/prof_$TAG"
