#!/bin/bash
# A/B of the mirror's host-thread -> stream mapping (troy.cpp current_stream(), TROY_STREAMS): the reference tool's -c N mode through
# tests/cpp/he_bench_driver for one stream per host thread (rounds 1-5) and the bounded stream set.  tools/streams_ab.sh [modes...] -> stdout
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
MODES="${@:-per-thread 1 2 4 8}"
for m in $MODES; do
  echo "== TROY_STREAMS=$m"
  TROY_STREAMS=$m "$ROOT/tests/cpp/he_bench_driver" bench 10 2>&1 | grep -E "^single_(threads[0-9]+_)?(three_calls|fused)(_us_per_op|_ops_per_s) |^OK|FAIL|EXCEPTION"
done
