#!/bin/bash
# Kernel trace of the cfg4 workload (BFV N = 32768 L = 10 multiply + relinearize, 256 ops in chunks of 64) -> gpurun_out/<tag>_summary.txt
set -e
TAG=${1:-r06_cfg4}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload cfg4 --total 256 --steps 5 --warmup 2 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats -d "$OUT/prof_$TAG/trace" -o bench -- python3 $ARGS > "$OUT/${TAG}_bench.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d "$OUT/prof_$TAG/$C" -o bench -- python3 $ARGS > /dev/null 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace -d "$OUT/prof_$TAG/SQ" -o bench -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d "$OUT/prof_$TAG/GRBM" -o bench -- python3 $ARGS > /dev/null 2>&1
cd "$ROOT"
python3 tools/rocpd_summary.py "$OUT/prof_$TAG/trace/bench_results.db" --pmc "$OUT/prof_$TAG/FETCH_SIZE/bench_results.db" --pmc "$OUT/prof_$TAG/WRITE_SIZE/bench_results.db" \
        --pmc "$OUT/prof_$TAG/SQ/bench_results.db" --pmc "$OUT/prof_$TAG/GRBM/bench_results.db" > "$OUT/${TAG}_summary.txt"
# counter bytes of the two dominant launches (bench.py attaches them to the cfg4 line as roofline.traffic; sha-locked like the headline's record)
python3 tools/collect_counters.py "ksmac2_kernel<15" 786432 64 "$OUT/${TAG}_ksmac_counters.json" \
        "$OUT/prof_$TAG/FETCH_SIZE/bench_results.db" "$OUT/prof_$TAG/WRITE_SIZE/bench_results.db" "$OUT/prof_$TAG/SQ/bench_results.db" "$OUT/prof_$TAG/GRBM/bench_results.db"
python3 tools/collect_counters.py "tensor_core_kernel" 3145728 64 "$OUT/${TAG}_tensor_bsk_counters.json" \
        "$OUT/prof_$TAG/FETCH_SIZE/bench_results.db" "$OUT/prof_$TAG/WRITE_SIZE/bench_results.db" "$OUT/prof_$TAG/SQ/bench_results.db" "$OUT/prof_$TAG/GRBM/bench_results.db"
python3 tools/collect_counters.py "tensor_core_kernel" 2621440 64 "$OUT/${TAG}_tensor_q_counters.json" \
        "$OUT/prof_$TAG/FETCH_SIZE/bench_results.db" "$OUT/prof_$TAG/WRITE_SIZE/bench_results.db" "$OUT/prof_$TAG/SQ/bench_results.db" "$OUT/prof_$TAG/GRBM/bench_results.db"
python3 tools/collect_valu.py "$OUT/${TAG}_valu.json" "$OUT/prof_$TAG/SQ/bench_results.db" "$OUT/prof_$TAG/GRBM/bench_results.db" --units 64 --unit-name "ciphertext pairs (one chunk)" --base-calls 28
tail -2 "$OUT/${TAG}_bench.log"
# the databases are scratch (tens of MB each; gpurun copies back at most 64 MiB): the summaries above are what profiles/ keeps
rm -rf "$OUT/prof_$TAG"
