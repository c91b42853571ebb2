#!/bin/bash
# PMC pass (issue utilisation) of the mixed-chain relinearize kernels -> gpurun_out/<tag>_mixed_pmc.txt
set -e
TAG=${1:-mixed}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace -d "$OUT/prof_$TAG/SQ" -o mixed -- python3 $ROOT/tools/profile_mixed.py > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/prof_$TAG/GRBM" -o mixed -- python3 $ROOT/tools/profile_mixed.py > /dev/null 2>&1
cd "$ROOT"
python3 tools/rocpd_summary.py "$OUT/prof_$TAG/SQ/mixed_results.db" --pmc "$OUT/prof_$TAG/SQ/mixed_results.db" --pmc "$OUT/prof_$TAG/GRBM/mixed_results.db" > "$OUT/${TAG}_mixed_pmc.txt"
rm -rf "$OUT/prof_$TAG"
