#!/bin/bash
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export LD_LIBRARY_PATH=$ROOT/troy-nova_amd:$ROOT/troy-nova_amd/troy:$LD_LIBRARY_PATH
rocprofv3 --kernel-trace --stats -d "$OUT/single_tr" -o single -- $ROOT/tests/cpp/he_bench_driver single > "$OUT/single_run.txt" 2>&1
python3 $ROOT/tools/rocpd_summary.py "$OUT/single_tr/single_results.db" | head -40 | cut -c1-170
tail -3 "$OUT/single_run.txt"
