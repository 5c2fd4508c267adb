# Builds everything in-tree (the .so files travel to the GPU box with the snapshot).
#   make            HIP library + pybind11 modules + CPU oracle
#   make lib        envpool_amd/libenvpool_amd.so   (hipcc, gfx950)
#   make pymod      envpool_amd/dummy/dummy_envpool*.so, envpool_amd/ygopro/ygopro_envpool*.so
#   make oracle     oracle/liboracle.so             (gcc; test infrastructure)
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
PYTHON ?= python3
ARCH ?= gfx950
CLANG ?= /opt/rocm/lib/llvm/bin/clang
EXT := $(shell $(PYTHON) -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
PYINC := $(shell $(PYTHON) -m pybind11 --includes)

CSRC := envpool_amd/csrc
LIB := envpool_amd/libenvpool_amd.so
HIP_SRCS := $(wildcard $(CSRC)/*.hip)
HIP_DEPS := $(wildcard $(CSRC)/*.hpp) $(wildcard $(CSRC)/*.inl) $(wildcard include/*.h)
PYMODS := envpool_amd/dummy/dummy_envpool$(EXT) $(if $(wildcard $(CSRC)/py_ygopro.cpp),envpool_amd/ygopro/ygopro_envpool$(EXT))

all: lib pymod oracle synth examples

lib: $(LIB)

$(LIB): $(HIP_SRCS) $(HIP_DEPS)
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -Iinclude -o $@ $(HIP_SRCS)

pymod: $(PYMODS)

envpool_amd/dummy/dummy_envpool$(EXT): $(CSRC)/py_dummy.cpp $(CSRC)/py_common.h include/envpool_amd.h $(LIB)
	$(CXX) -O2 -std=c++17 -fPIC -shared -fvisibility=hidden $(PYINC) -Iinclude -o $@ $< \
	  -Lenvpool_amd -lenvpool_amd -Wl,-rpath,'$$ORIGIN/..'

envpool_amd/ygopro/ygopro_envpool$(EXT): $(CSRC)/py_ygopro.cpp $(CSRC)/py_common.h include/envpool_amd.h include/envpool_amd_ygo.h $(LIB)
	$(CXX) -O2 -std=c++17 -fPIC -shared -fvisibility=hidden $(PYINC) -Iinclude -o $@ $< \
	  -Lenvpool_amd -lenvpool_amd -Wl,-rpath,'$$ORIGIN/..'

oracle:
	$(MAKE) -C oracle

synth: tools/libygosynth.so

tools/libygosynth.so: tools/ygo_synth.c include/envpool_amd_ygo.h include/envpool_amd.h
	gcc -O2 -fPIC -shared -std=gnu11 -Wall -pthread -o $@ $<

# Diagnostic programs (launch floors, host-side latencies of the step loop). Not part of `all`.
# the C ABI from plain C: gcc only, no HIP headers
examples: examples/dummy_step_loop examples/ygo_pool_loop
examples/ygo_pool_loop: examples/ygo_pool_loop.c include/envpool_amd.h include/envpool_amd_ygo.h $(LIB)
	gcc -O2 -std=c11 -Wall -Iinclude $< -Lenvpool_amd -lenvpool_amd -Wl,-rpath,'$$ORIGIN/../envpool_amd' -o $@
examples/dummy_step_loop: examples/dummy_step_loop.c include/envpool_amd.h $(LIB)
	gcc -O2 -std=c11 -Wall -Iinclude $< -Lenvpool_amd -lenvpool_amd -Wl,-rpath,'$$ORIGIN/../envpool_amd' -o $@

diag: tools/launch_floor tools/host_latency tools/doorbell_floor tools/server_ab tools/pcie_write_ceiling
tools/pcie_write_ceiling: tools/pcie_write_ceiling.cpp
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 -o $@ $<
tools/server_ab: tools/server_ab.cpp $(LIB)
	$(HIPCC) -O2 -std=c++17 -Iinclude $< -Lenvpool_amd -lenvpool_amd -Wl,-rpath,'$$ORIGIN/../envpool_amd' -o $@
tools/doorbell_floor: tools/doorbell_floor.cpp
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 -o $@ $<
tools/launch_floor: tools/launch_floor.cpp
	$(HIPCC) --offload-arch=$(ARCH) -O2 -o $@ $<
tools/host_latency: tools/host_latency.cpp $(LIB)
	$(HIPCC) -O2 -std=c++17 -Iinclude $< -Lenvpool_amd -lenvpool_amd -Wl,-rpath,'$$ORIGIN/../envpool_amd' -o $@

# ThreadSanitizer over the host logic of ep_ygo_pool (ygo_pool.hip compiled as plain host C++
# against a mock encoder: TSan and the ROCm runtime do not coexist). Not part of `all`.
tsan-pool: build/tsan_ygo_pool
	TSAN_OPTIONS=halt_on_error=1 build/tsan_ygo_pool
build/tsan_ygo_pool: tools/tsan_ygo_pool.cpp $(CSRC)/ygo_pool.hip $(CSRC)/common.hpp include/envpool_amd_ygo.h include/envpool_amd.h
	mkdir -p build
	$(CLANG)++ -std=c++17 -O1 -g -fsanitize=thread -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude \
	  -o $@ tools/tsan_ygo_pool.cpp $(CSRC)/ygo_pool.hip -pthread

# Host-code sanitizer target (GPU sanitizers are not available on this pool): the library's host
# code under ASan + UBSan, driven by tools/sanitize_host.cpp on the CPU. Not part of `all`.
SANFLAGS := -fsanitize=address -fsanitize=undefined -fno-omit-frame-pointer
sanitize: build/sanitize_host
	ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 build/sanitize_host

build/sanitize_host: tools/sanitize_host.cpp tools/ygo_synth.c $(HIP_SRCS) $(HIP_DEPS)
	mkdir -p build
	$(CLANG) -O1 -g -std=gnu11 -pthread $(SANFLAGS) -c -o build/ygo_synth_san.o tools/ygo_synth.c
	$(CLANG)++ -O1 -g -std=c++17 $(SANFLAGS) -c -o build/sanitize_host.o tools/sanitize_host.cpp
	for f in $(HIP_SRCS); do $(HIPCC) --offload-arch=$(ARCH) -O1 -g -std=c++17 -fPIC -Iinclude \
	  $(foreach f,$(SANFLAGS),-Xarch_host $(f)) -c -o build/san_$$(basename $$f .hip).o $$f || exit 1; done
	$(HIPCC) --offload-arch=$(ARCH) $(SANFLAGS) -o $@ build/sanitize_host.o \
	  $(patsubst $(CSRC)/%.hip,build/san_%.o,$(HIP_SRCS)) build/ygo_synth_san.o -pthread

clean:
	rm -f $(LIB) $(PYMODS) tools/libygosynth.so examples/dummy_step_loop examples/ygo_pool_loop tools/launch_floor tools/host_latency tools/doorbell_floor tools/server_ab tools/pcie_write_ceiling
	rm -rf build
	$(MAKE) -C oracle clean

.PHONY: all lib pymod oracle synth examples sanitize tsan-pool diag clean
