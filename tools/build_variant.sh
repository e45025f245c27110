#!/bin/bash
# Build an A/B variant of libptgpu.so with extra compiler flags:  tools/build_variant.sh <name> <flags...>
# -> build/variants/libptgpu_<name>.so ; select it at run time with PT_GPU_LIB=<path>.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/variants
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -Iinclude -Ipath-tracer_amd/csrc -Wall "$@" \
    -c path-tracer_amd/csrc/pt_gpu.hip -o build/variants/pt_gpu_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libptgpu_$name.so build/variants/pt_gpu_$name.o build/host/*.o -lz -pthread
echo build/variants/libptgpu_$name.so
