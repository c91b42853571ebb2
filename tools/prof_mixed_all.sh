#!/bin/bash
# kernel trace + SQ counters of relinearize (and the fused chain) on chains with moduli >= 2^50 -> gpurun_out/<tag>_mixed_{trace,pmc}_<shape>.txt
TAG=${1:-r06}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export MIXED_CHAIN=1
for SH in "8192:60,40,40,60:3:1024" "16384:60,50,50,50,50,60:5:512"; do
  NAME=$(echo $SH | cut -d: -f1)
  export MIXED_SHAPE=$SH
  rocprofv3 --kernel-trace --stats -d "$OUT/prof_$TAG/trace$NAME" -o mixed -- python3 $ROOT/tools/profile_mixed.py > "$OUT/${TAG}_mixed_$NAME.log" 2>&1
  python3 $ROOT/tools/rocpd_summary.py "$OUT/prof_$TAG/trace$NAME/mixed_results.db" > "$OUT/${TAG}_mixed_trace_$NAME.txt"
  if [ "$2" == "pmc" ]; then
    rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace -d "$OUT/prof_$TAG/SQ$NAME" -o mixed -- python3 $ROOT/tools/profile_mixed.py > /dev/null 2>&1
    python3 $ROOT/tools/rocpd_summary.py "$OUT/prof_$TAG/SQ$NAME/mixed_results.db" --pmc "$OUT/prof_$TAG/SQ$NAME/mixed_results.db" > "$OUT/${TAG}_mixed_pmc_$NAME.txt"
  fi
done
rm -rf "$OUT/prof_$TAG"
