#!/bin/bash
# As tools/ksmac_variants.sh, for troyn_ksmaci.o (compile-time knobs of ksmaci_kernel), timed with tools/bench_mixed.py.  Development tool.
#   tools/ksmaci_variants.sh build "NAME:-DFLAG ..." ...      tools/ksmaci_variants.sh run
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CS="$ROOT/troy-nova_amd/csrc"; AB="$ROOT/troy-nova_amd/ablate"
mkdir -p "$AB"
if [ "$1" == "build" ]; then
  shift
  for spec in "$@"; do
    name="${spec%%:*}"; flags="${spec#*:}"
    ( cd "$CS" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wall -Wno-unused-function -fPIC -ffp-contract=off $flags -c -o "$AB/ksmaci_$name.o" troyn_ksmaci.hip \
      && /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o "$AB/libtroyni_$name.so" troyn.o troyn_ntt_f64_small.o troyn_ntt_f64_large.o troyn_ntt_u64_small.o troyn_ntt_u64_large.o troyn_ksmac2.o "$AB/ksmaci_$name.o" troyn_behz2.o ) &
  done
  wait
  ls "$AB"/libtroyni_*.so
else
  cd "$ROOT"
  for round in 1 2; do
    for lib in "" $(ls "$AB"/libtroyni_*.so 2>/dev/null); do
      name=$(basename "${lib:-default}")
      if [ -n "$lib" ]; then export TROYN_LIB="$lib"; else unset TROYN_LIB; fi
      python tools/bench_mixed.py 2>/dev/null | python3 -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('$name', d['n'], d['chain'], 'relin', d['relinearize_ops_per_s'], 'inner_ms', d.get('inner_product_ms'), 'chain', d.get('ckks_mul_relin_rescale_ops_per_s'))"
    done
  done
fi
