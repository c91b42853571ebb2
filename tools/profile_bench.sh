#!/bin/bash
# Profiles the default bench.py run on the GPU box (rocprofv3 kernel trace + separate PMC passes) and writes the summaries that
# profiles/ keeps:  tools/profile_bench.sh <tag>   ->  gpurun_out/<tag>_summary.txt, gpurun_out/<tag>_ksmac_counters.json
# Run through gpurun from the repository root:  gpurun -- 'bash tools/profile_bench.sh r02_bench_v1'
set -e
TAG=${1:-r06_bench}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 10 --warmup 1 --inner 4 --no-cpu-baseline --no-extra"
# the fused chain on ONE stream (the library's default) for the kernel trace and the counter passes
export TROYN_MRR_CHUNK=0
rocprofv3 --kernel-trace --stats -d "$OUT/prof_$TAG/trace" -o bench -- python3 $ARGS > "$OUT/${TAG}_bench.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d "$OUT/prof_$TAG/$C" -o bench -- python3 $ARGS > /dev/null 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace -d "$OUT/prof_$TAG/SQ" -o bench -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d "$OUT/prof_$TAG/GRBM" -o bench -- python3 $ARGS > /dev/null 2>&1
cd "$ROOT"
python3 tools/rocpd_summary.py "$OUT/prof_$TAG/trace/bench_results.db" --pmc "$OUT/prof_$TAG/FETCH_SIZE/bench_results.db" --pmc "$OUT/prof_$TAG/WRITE_SIZE/bench_results.db" \
        --pmc "$OUT/prof_$TAG/SQ/bench_results.db" --pmc "$OUT/prof_$TAG/GRBM/bench_results.db" > "$OUT/${TAG}_summary.txt"
python3 tools/collect_counters.py "ksmac2_kernel<14, true, 0, false, true" $((1024 * 6 * 2 * 256)) 1024 "$OUT/${TAG}_ksmac_counters.json" \
        "$OUT/prof_$TAG/FETCH_SIZE/bench_results.db" "$OUT/prof_$TAG/WRITE_SIZE/bench_results.db" "$OUT/prof_$TAG/SQ/bench_results.db" "$OUT/prof_$TAG/GRBM/bench_results.db"
# vector-ALU issue accounting of every kernel of the pass (the chain-wide FP64 floor bench.py reports as roofline.floor)
python3 tools/collect_valu.py "$OUT/${TAG}_chain_valu.json" "$OUT/prof_$TAG/SQ/bench_results.db" "$OUT/prof_$TAG/GRBM/bench_results.db" --units 1024 --unit-name "ciphertext pairs" \
        --pass "ksmac2_kernel<14, true, 0, false, true=1" --pass "false, true, true, 5, true>=1" --pass "true, true, true, 3, false>=1" --pass "true, true, true, 4, false>=1" \
        --pass "true, true, true, 0, true>@2097152=1" --pass "ksmac_prepare_keys_kernel=1"
# the chunked option (two halves on two internal streams), kernel trace only, for the record
export TROYN_MRR_CHUNK=512
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/prof_$TAG/trace2" -o bench -- python3 $ARGS > "$OUT/${TAG}_bench_two_streams.log" 2>&1
cd "$ROOT"
python3 tools/rocpd_summary.py "$OUT/prof_$TAG/trace2/bench_results.db" > "$OUT/${TAG}_two_streams_summary.txt"
tail -3 "$OUT/${TAG}_bench.log"
# the databases are scratch (tens of MB each; gpurun copies back at most 64 MiB): the summaries above are what profiles/ keeps
rm -rf "$OUT/prof_$TAG"
