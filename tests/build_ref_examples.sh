#!/bin/bash
# Builds the reference's OWN example programs (/root/reference/examples/*.cu, compiled where they lie -- nothing is copied into the
# repo) against this repository's host-side mirror (troy-nova_amd/troy/*.h, libtroy_amd.so) into tests/_ref_examples/ref_examples.
# Purpose: the drop-in check of SURVEY 8b -- a program written for the reference compiles unchanged against the mirror and, on the
# GPU box, runs on the HIP path (tests/test_gpu_ref_examples.py).  Test infrastructure only; needs /root/reference, so it runs in the
# build container (the GPU box uses the prebuilt binary).  This is NOT a build of the reference library (that would
# live under oracle/_ref and needs the CUDA toolkit): only the reference's example CALLERS, linked against this implementation.
#   * the examples include "examples.h", which includes "../src/troy.h": the sources are fed to the compiler on stdin so that the
#     include resolves to a generated header (this mirror's headers + the helper functions of the reference's examples.h) in the
#     scratch build directory; the generated header is not kept.
#   * 30_issue_multithread.cu is a CUDA programming demonstration (its own __global__ kernels and <<<>>> launches on device arrays of
#     the reference's internal utils/box.h), not a user of the library API; it is replaced by a stub.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
REF=${REF:-/root/reference}
[ -d "$REF/examples" ] || { echo "no reference tree at $REF: nothing to build"; exit 0; }
BUILD="$(mktemp -d "${TMPDIR:-/tmp}/ref_examples.XXXXXX")"
trap 'rm -rf "$BUILD"' EXIT
PKG="$ROOT/troy-nova_amd"
{
  echo '#pragma once'
  for h in conv2d.h matmul.h ring2k.h troy.h; do echo "#include \"$PKG/troy/$h\""; done
  echo '#include <future>'
  grep -v '#include "../src/troy.h"' "$REF/examples/examples.h" | grep -v '#pragma once'
} > "$BUILD/examples.h"
CXXFLAGS="-O1 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$BUILD -w"
OBJS=""
cd "$BUILD"
for f in "$REF"/examples/*.cu; do
  b="$(basename "$f" .cu)"
  [ "$b" = "30_issue_multithread" ] && continue
  g++ $CXXFLAGS -x c++ -c -o "$BUILD/$b.o" - < "$f"
  OBJS="$OBJS $BUILD/$b.o"
done
printf '#include <iostream>\nvoid example_issue_multithread() { std::cout << "skipped: a CUDA kernel-launch demonstration, not a user of the library API" << std::endl; }\n' > "$BUILD/stub.cpp"
g++ $CXXFLAGS -c -o "$BUILD/stub.o" "$BUILD/stub.cpp"
mkdir -p "$ROOT/tests/_ref_examples"
g++ -o "$ROOT/tests/_ref_examples/ref_examples" $OBJS "$BUILD/stub.o" -L"$PKG" -ltroy_amd -ltroyn -L/opt/rocm/lib -lamdhip64 -lpthread \
    -Wl,-rpath,'$ORIGIN/../../troy-nova_amd' -Wl,-rpath,/opt/rocm/lib
echo "built tests/_ref_examples/ref_examples"
