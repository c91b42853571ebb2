#!/bin/bash
# Kernel traces of the cfg4 workload under both BEHZ auxiliary bases (TROYN_BEHZ_BASE=ref: reference 61-bit / default: primes below 2^50) -> gpurun_out/<tag>_{ref,small}_summary.txt
set -e
TAG=${1:-r04_cfg4}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload cfg4 --total 256 --steps 5 --warmup 2 --no-cpu-baseline --no-extra"
for BASE in ref small; do
  if [ "$BASE" = small ]; then unset TROYN_BEHZ_BASE; else export TROYN_BEHZ_BASE=ref; fi
  rocprofv3 --kernel-trace --stats -d "$OUT/prof_${TAG}_$BASE/trace" -o bench -- python3 $ARGS > "$OUT/${TAG}_${BASE}_bench.log" 2>&1
  (cd "$ROOT" && python3 tools/rocpd_summary.py "$OUT/prof_${TAG}_$BASE/trace/bench_results.db" > "$OUT/${TAG}_${BASE}_summary.txt")
  tail -1 "$OUT/${TAG}_${BASE}_bench.log" | cut -c1-200
  rm -rf "$OUT/prof_${TAG}_$BASE"
done
