# Builds everything in-tree (the .so files travel to the GPU box with the snapshot).
#   make            HIP library + pybind11 modules + CPU oracle
#   make lib        envpool_amd/libenvpool_amd.so   (hipcc, gfx950)
#   make pymod      envpool_amd/dummy/dummy_envpool*.so, envpool_amd/ygopro/ygopro_envpool*.so
#   make oracle     oracle/liboracle.so             (gcc; test infrastructure)
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
PYTHON ?= python3
ARCH ?= gfx950
EXT := $(shell $(PYTHON) -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
PYINC := $(shell $(PYTHON) -m pybind11 --includes)

CSRC := envpool_amd/csrc
LIB := envpool_amd/libenvpool_amd.so
HIP_SRCS := $(wildcard $(CSRC)/*.hip)
HIP_DEPS := $(wildcard $(CSRC)/*.hpp) $(wildcard $(CSRC)/*.inl) $(wildcard include/*.h)
PYMODS := envpool_amd/dummy/dummy_envpool$(EXT) $(if $(wildcard $(CSRC)/py_ygopro.cpp),envpool_amd/ygopro/ygopro_envpool$(EXT))

all: lib pymod oracle synth

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

clean:
	rm -f $(LIB) $(PYMODS) tools/libygosynth.so
	$(MAKE) -C oracle clean

.PHONY: all lib pymod oracle synth clean
