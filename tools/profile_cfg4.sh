#!/bin/bash
# Profiles BASELINE config 4 (BFV N=32768 L=10, batch 64: BEHZ multiply, relinearize) on the GPU box: rocprofv3 kernel trace plus
# separate FETCH_SIZE / WRITE_SIZE passes of tools/bench_configs.py --only cfg4.
#   gpurun -- 'bash tools/profile_cfg4.sh r02_cfg4'   ->  gpurun_out/<tag>_summary.txt, gpurun_out/<tag>.json
set -e
TAG=${1:-r02_cfg4}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/tools/bench_configs.py --only cfg4 --reps 4"
rocprofv3 --kernel-trace --stats -d "$OUT/prof_$TAG/trace" -o cfg4 -- python3 $ARGS > "$OUT/${TAG}.json" 2> "$OUT/${TAG}.log"
if [ "$2" != "--no-pmc" ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace -d "$OUT/prof_$TAG/$C" -o cfg4 -- python3 $ARGS > /dev/null 2>&1
  done
  PMC="--pmc $OUT/prof_$TAG/FETCH_SIZE/cfg4_results.db --pmc $OUT/prof_$TAG/WRITE_SIZE/cfg4_results.db"
fi
cd "$ROOT"
python3 tools/rocpd_summary.py "$OUT/prof_$TAG/trace/cfg4_results.db" $PMC > "$OUT/${TAG}_summary.txt"
cat "$OUT/${TAG}.json"
