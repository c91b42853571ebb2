#!/bin/bash
# Counters of the transform alone at N = 16384 (tools/ntt_alone.py 14): kernel trace + FETCH/WRITE + SQ issue counters -> gpurun_out/<tag>_summary.txt
set -e
TAG=${1:-r06_ntt}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/tools/ntt_alone.py ${2:-14}"
rocprofv3 --kernel-trace --stats -d "$OUT/prof_$TAG/trace" -o bench -- python3 $ARGS > "$OUT/${TAG}.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d "$OUT/prof_$TAG/$C" -o bench -- python3 $ARGS > /dev/null 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace -d "$OUT/prof_$TAG/SQ" -o bench -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d "$OUT/prof_$TAG/GRBM" -o bench -- python3 $ARGS > /dev/null 2>&1
cd "$ROOT"
python3 tools/rocpd_summary.py "$OUT/prof_$TAG/trace/bench_results.db" --pmc "$OUT/prof_$TAG/FETCH_SIZE/bench_results.db" --pmc "$OUT/prof_$TAG/WRITE_SIZE/bench_results.db" \
        --pmc "$OUT/prof_$TAG/SQ/bench_results.db" --pmc "$OUT/prof_$TAG/GRBM/bench_results.db" > "$OUT/${TAG}_summary.txt"
grep "^{" "$OUT/${TAG}.log" || true
rm -rf "$OUT/prof_$TAG"
