#!/bin/bash
# Builds the reference's OWN test suite (/root/reference/test/*.cu and test/app/*.cu: 362 googletest cases; compiled where they lie -- nothing of the reference is
# copied into the repository) against this repository's host-side mirror (troy-nova_amd/troy/*.h, libtroy_amd.so) into tests/_ref_tests/ref_tests.
# Purpose: the drop-in check of SURVEY 8b taken to its end -- the tests the reference's maintainers wrote pass on the HIP path through the mirror
# (tests/test_gpu_ref_tests.py runs the Device* cases; the mirror has no host path, so Host* cases do not apply).  Test infrastructure only; needs /root/reference,
# so it runs in the build container (the GPU box uses the prebuilt binary, which is git-ignored like tests/_ref_examples).  This is NOT a build of the reference
# library: only the reference's test CALLERS, linked against this implementation.
#   * the sources include "../src/<name>.h": they are compiled in a scratch directory where those resolve to one-line forwarders to the mirror's headers;
#   * <gtest/gtest.h> (an un-vendored submodule of the reference) resolves to tests/ref_tests_support/gtest/gtest.h, a 50-line runner written for this repository;
#   * "cuda_runtime.h" (test_adv.h:3) resolves to <hip/hip_runtime.h> plus the four names the tests spell out (cudaError_t, cudaSuccess, cudaGetDeviceCount,
#     cudaDeviceSynchronize, cudaStreamSynchronize): the tests ask the runtime how many devices there are;
#   * not built: test/utils/*.cu and test/modulus.cu (unit tests of the reference's internal utilities and CUDA kernels, not users of the public API),
#     (test/serialize_zstd.cu IS built: the mirror loads the zstd runtime library, troy.cpp).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
REF=${REF:-/root/reference}
[ -d "$REF/test" ] || { echo "no reference tree at $REF: nothing to build"; exit 0; }
W="$(mktemp -d "${TMPDIR:-/tmp}/ref_tests.XXXXXX")"
trap 'rm -rf "$W"' EXIT
PKG="$ROOT/troy-nova_amd"
T="$PKG/troy"
mkdir -p "$W/src/app" "$W/src/utils" "$W/test/app"
for h in he_context batch_encoder ckks_encoder evaluator encryptor key_generator decryptor batch_utils troy lwe_ciphertext; do echo "#include \"$T/troy.h\"" > "$W/src/$h.h"; done
echo "#include \"$T/troy.h\"" > "$W/src/utils/box.h"
echo "#include \"$T/bench_timer.h\"" > "$W/src/utils/timer.h"
for h in bfv_ring2k matmul conv2d cipher2d encoder_adapter; do printf '#include "%s/troy.h"\n#include "%s/ring2k.h"\n#include "%s/matmul.h"\n#include "%s/conv2d.h"\n' "$T" "$T" "$T" "$T" > "$W/src/app/$h.h"; done
printf '#include <hip/hip_runtime.h>\n#define cudaError_t hipError_t\n#define cudaSuccess hipSuccess\n#define cudaGetDeviceCount hipGetDeviceCount\n#define cudaDeviceSynchronize hipDeviceSynchronize\n' > "$W/test/cuda_runtime.h"
# the bench tool waits with cudaStreamSynchronize(0): under the reference's --default-stream per-thread build that is the CALLING THREAD's stream, i.e. the mirror's
# utils::stream_sync() (a literal hipStreamSynchronize(0) would wait for every thread's stream)
printf 'namespace troy { namespace utils { void stream_sync(); } }\n#define cudaStreamSynchronize(s) ((s) == 0 ? (troy::utils::stream_sync(), hipSuccess) : hipStreamSynchronize(s))\n' >> "$W/test/cuda_runtime.h"
cp "$REF"/test/*.h "$REF"/test/*.cu "$W/test/"
cp "$REF"/test/app/*.cu "$W/test/app/"
CXXFLAGS="-O1 -std=c++17 -w -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$ROOT/tests/ref_tests_support -I."
cd "$W/test"
OBJS=""
for f in test_adv test_multithread evaluator evaluator_batched encryptor encryptor_batched serialize serialize_zstd lwe batch_encoder batch_encoder_batched ckks_encoder he_context \
         special_prime_for_encryption multithread app/matmul app/conv2d app/matmul_ckks app/conv2d_ckks app/bfv_ring2k app/matmul_ring2k app/conv2d_ring2k; do
  o="$W/$(echo "$f" | tr / _).o"
  g++ $CXXFLAGS -x c++ -c -o "$o" "$f.cu" &
  OBJS="$OBJS $o"
  while [ "$(jobs -r | wc -l)" -ge 6 ]; do sleep 0.2; done
done
wait
g++ $CXXFLAGS -c -o "$W/main.o" "$ROOT/tests/ref_tests_support/ref_tests_main.cpp"
mkdir -p "$ROOT/tests/_ref_tests"
g++ -o "$ROOT/tests/_ref_tests/ref_tests" $OBJS "$W/main.o" -L"$PKG" -ltroy_amd -ltroyn -L/opt/rocm/lib -lamdhip64 -lpthread \
    -Wl,-rpath,'$ORIGIN/../../troy-nova_amd' -Wl,-rpath,/opt/rocm/lib
echo "built tests/_ref_tests/ref_tests ($("$ROOT/tests/_ref_tests/ref_tests" --list | wc -l) cases)"
# the reference's bench tool (test/bench/he_operations.cu = `troybench`, its own main) against the mirror: tests/_ref_tests/ref_troybench -D ... is the tool a
# maintainer of the reference would run (tests/cpp/he_bench_driver.cpp is this repository's counterpart, which also times the batched and fused entries)
mkdir -p "$W/test/bench"
cp "$REF"/test/bench/he_operations.cu "$REF"/test/bench/argument_helper.h "$W/test/bench/"
cp "$REF"/test/argparse.cpp "$W/test/"
( cd "$W/test/bench" && g++ $CXXFLAGS -I.. -x c++ -c -o "$W/troybench.o" he_operations.cu )
g++ $CXXFLAGS -c -o "$W/argparse.o" "$W/test/argparse.cpp"
g++ -o "$ROOT/tests/_ref_tests/ref_troybench" "$W/troybench.o" "$W/argparse.o" "$W/test_adv.o" "$W/test_multithread.o" -L"$PKG" -ltroy_amd -ltroyn -L/opt/rocm/lib -lamdhip64 -lpthread \
    -Wl,-rpath,'$ORIGIN/../../troy-nova_amd' -Wl,-rpath,/opt/rocm/lib
echo "built tests/_ref_tests/ref_troybench"
# ... and its matmul / conv2d bench tools (test/bench/matmul.cu, conv2d.cu: BASELINE config 5 is `bench_matmul -D --bfv -m 512 -r 512 -n 512 ...`)
cp "$REF"/test/bench/matmul.cu "$REF"/test/bench/conv2d.cu "$W/test/bench/"
for t in matmul conv2d; do
  ( cd "$W/test/bench" && g++ $CXXFLAGS -I.. -x c++ -c -o "$W/bench_$t.o" $t.cu )
  g++ -o "$ROOT/tests/_ref_tests/ref_bench_$t" "$W/bench_$t.o" "$W/argparse.o" "$W/test_adv.o" "$W/test_multithread.o" -L"$PKG" -ltroy_amd -ltroyn -L/opt/rocm/lib -lamdhip64 -lpthread \
      -Wl,-rpath,'$ORIGIN/../../troy-nova_amd' -Wl,-rpath,/opt/rocm/lib
  echo "built tests/_ref_tests/ref_bench_$t"
done
