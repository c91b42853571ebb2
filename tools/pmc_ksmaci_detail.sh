#!/bin/bash
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export MIXED_SHAPE="8192:60,40,40,60:3:1024"
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU --kernel-trace -d "$OUT/p2a" -o mixed -- python3 $ROOT/tools/profile_mixed.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM --kernel-trace -d "$OUT/p2b" -o mixed -- python3 $ROOT/tools/profile_mixed.py > /dev/null 2>&1
python3 $ROOT/tools/rocpd_summary.py "$OUT/p2a/mixed_results.db" --pmc "$OUT/p2a/mixed_results.db" --pmc "$OUT/p2b/mixed_results.db" | grep "ksmaci_kernel<13, 1>\|ksmac2_kernel<13, false, 0, true, false, true" | cut -c1-40,118-200
rm -rf "$OUT/p2a" "$OUT/p2b"
