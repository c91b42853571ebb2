#!/bin/bash
# Builds variant copies of libtroyn.so that differ only in troyn_ksmac2.o (compile-time knobs of ksmac2_kernel) -- run HERE (cross-compile), then
# `tools/ksmac_variants.sh run` on the GPU box times the headline with each (bench.py --no-extra) in one session.  Development tool.
#   tools/ksmac_variants.sh build "NAME:-DFLAG -DFLAG2" ...      tools/ksmac_variants.sh run [steps]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CS="$ROOT/troy-nova_amd/csrc"; AB="$ROOT/troy-nova_amd/ablate"
mkdir -p "$AB"
if [ "$1" == "build" ]; then
  shift
  for spec in "$@"; do
    name="${spec%%:*}"; flags="${spec#*:}"
    ( cd "$CS" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wall -Wno-unused-function -fPIC -ffp-contract=off $flags -c -o "$AB/ksmac2_$name.o" troyn_ksmac2.hip \
      && /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o "$AB/libtroyn_$name.so" troyn.o troyn_ntt_f64_small.o troyn_ntt_f64_large.o troyn_ntt_u64_small.o troyn_ntt_u64_large.o "$AB/ksmac2_$name.o" troyn_ksmaci.o troyn_behz2.o ) &
  done
  wait
  ls -la "$AB"/*.so
else
  STEPS=${2:-10}
  cd "$ROOT"
  for round in 1 2; do
    for lib in "" $(ls "$AB"/libtroyn_*.so 2>/dev/null); do
      name=$(basename "${lib:-default}")
      if [ -n "$lib" ]; then export TROYN_LIB="$lib"; else unset TROYN_LIB; fi
      python bench.py --steps $STEPS --warmup 2 --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']   # the headline is the LAST line
print('$name', 'ops/s', d['value'], 'ksmac_launch_ms', r.get('launch_ms'), 'frac', r['frac'])"
    done
  done
fi
