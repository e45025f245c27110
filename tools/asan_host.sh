#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side code (ISF loader, PNG codec, profile parser, KD builder, scene
# generator, oracle): builds instrumented copies of libpthost.so / libptoracle.so, runs the CPU tests against
# them, restores the normal libraries.  (GPU sanitizers are not available on the pool; the HIP side is covered by
# the bit-exactness tests.)    usage: bash tools/asan_host.sh
set -eu
cd "$(dirname "$0")/.."
T=$(mktemp -d)
trap 'cp "$T/libpthost.orig.so" path-tracer_amd/libpthost.so; cp "$T/libptoracle.orig.so" oracle/libptoracle.so; rm -rf "$T"' EXIT
cp path-tracer_amd/libpthost.so "$T/libpthost.orig.so"
cp oracle/libptoracle.so "$T/libptoracle.orig.so"
SAN="-O1 -g -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer"
g++ -std=c++17 $SAN -Iinclude -pthread -shared -o path-tracer_amd/libpthost.so $(ls path-tracer_amd/host/*.cpp | grep -v cli_main) -lz
g++ -std=c++17 $SAN -ffp-contract=off -fopenmp -Iinclude -shared -o oracle/libptoracle.so oracle/pt_oracle.cpp
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider --deselect tests/test_distributed_gloo.py --deselect tests/test_cli.py
