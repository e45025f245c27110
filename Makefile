# Build everything in-tree:  make            (host lib, oracle, HIP library, CLI)
#                            make host|oracle|gpu|cli
# gfx950 only; hipcc cross-compiles without a GPU.
ROOT      := $(abspath .)
PKG       := path-tracer_amd
BUILD     := build
CXX       ?= g++
HIPCC     ?= /opt/rocm/bin/hipcc
CXXFLAGS  := -std=c++17 -O2 -fPIC -Wall -Wextra -Iinclude -pthread
# The integrator's f32 arithmetic must not be contracted into FMAs or
# re-associated: the reference is scalar Rust without fast-math.
ORACLE_FLAGS := -std=c++17 -O2 -fPIC -Wall -Wextra -ffp-contract=off -fno-fast-math -fopenmp -Iinclude
HIPFLAGS  := -std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -Iinclude -I$(PKG)/csrc -Wall

HOST_SRC  := $(wildcard $(PKG)/host/*.cpp)
HOST_LIB_SRC := $(filter-out $(PKG)/host/cli_main.cpp,$(HOST_SRC))
HOST_OBJ  := $(patsubst $(PKG)/host/%.cpp,$(BUILD)/host/%.o,$(HOST_LIB_SRC))

all: host oracle gpu cli

host: $(PKG)/libpthost.so
oracle: oracle/libptoracle.so
gpu: $(PKG)/libptgpu.so
cli: $(PKG)/path-tracer

$(BUILD)/host/%.o: $(PKG)/host/%.cpp include/ptgpu.h include/pthost.h $(PKG)/host/host_common.hpp
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS) -c $< -o $@

$(PKG)/libpthost.so: $(HOST_OBJ)
	$(CXX) -shared -o $@ $^ -lz -pthread

oracle/libptoracle.so: oracle/pt_oracle.cpp oracle/pt_oracle.h include/ptgpu.h
	$(CXX) $(ORACLE_FLAGS) -shared -o $@ oracle/pt_oracle.cpp

GPU_SRC := $(wildcard $(PKG)/csrc/*.hip)
GPU_HDR := $(wildcard $(PKG)/csrc/*.h) $(wildcard $(PKG)/csrc/*.hpp) include/ptgpu.h include/pthost.h
GPU_OBJ := $(patsubst $(PKG)/csrc/%.hip,$(BUILD)/gpu/%.o,$(GPU_SRC))
$(BUILD)/gpu/%.o: $(PKG)/csrc/%.hip $(GPU_HDR)
	@mkdir -p $(dir $@)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(PKG)/libptgpu.so: $(GPU_OBJ) $(HOST_OBJ)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(GPU_OBJ) $(HOST_OBJ) -lz -pthread -ldl

$(PKG)/path-tracer: $(PKG)/host/cli_main.cpp $(PKG)/libptgpu.so
	$(CXX) $(CXXFLAGS) -o $@ $< -L$(PKG) -lptgpu -Wl,-rpath,'$$ORIGIN'

clean:
	rm -rf $(BUILD) $(PKG)/*.so oracle/*.so $(PKG)/path-tracer oracle/_ref

.PHONY: all host oracle gpu cli clean
