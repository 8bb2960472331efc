# Builds everything native in-tree:
#   make lib      -> volrend_amd/libvolrend_hip.so   (gfx950 kernels + C ABI, hipcc)
#   make host     -> volrend_amd/libvolrend_host.a   (kept C++ host layer, g++)
#   make gather   -> volrend_amd/libvolrend_gather.so (the tile shard's RCCL collective, include/volrend_gather.h)
#   make cli      -> volrend_amd/bin/volrend_headless
#   make oracle   -> oracle/liboracle.so (+ oracle/_ref when /root/reference is mounted)
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
ROCM ?= /opt/rocm
PKG = volrend_amd
HOST = $(PKG)/csrc/host
HOST_SRC = $(HOST)/npz.cpp $(HOST)/n3tree.cpp $(HOST)/camera.cpp $(HOST)/opts.cpp \
           $(HOST)/imwrite.cpp $(HOST)/renderer.cpp $(HOST)/tile_shard.cpp $(HOST)/volume_renderer.cpp
HOST_OBJ = $(HOST_SRC:.cpp=.o)
CXXFLAGS = -O2 -std=c++17 -fPIC -Wall -Wextra -Iinclude

all: lib host gather cli

lib:
	python3 -m volrend_amd.build

# every object depends on every kept header: a layout change (RenderOptions, N3Tree) must never
# leave a stale object in the archive
HOST_HDR = $(wildcard include/*.h include/volrend/*.hpp include/volrend/internal/*.hpp)

$(HOST)/%.o: $(HOST)/%.cpp $(HOST_HDR)
	$(CXX) $(CXXFLAGS) -c $< -o $@

# the multi-GPU tile shard talks to the HIP runtime and RCCL directly
$(HOST)/tile_shard.o: $(HOST)/tile_shard.cpp $(HOST_HDR)
	$(CXX) $(CXXFLAGS) -I$(ROCM)/include -D__HIP_PLATFORM_AMD__ -c $< -o $@

# ... and so does the renderer facade (its frames are device memory)
$(HOST)/volume_renderer.o: $(HOST)/volume_renderer.cpp $(HOST_HDR)
	$(CXX) $(CXXFLAGS) -I$(ROCM)/include -D__HIP_PLATFORM_AMD__ -c $< -o $@

# the one translation unit that talks to RCCL: a shared library of its own (ctypes: bench.py --gpus N) that the
# CLI links as well
gather: $(PKG)/libvolrend_gather.so
$(PKG)/libvolrend_gather.so: $(HOST)/gather.cpp include/volrend_gather.h
	$(CXX) $(CXXFLAGS) -shared -I$(ROCM)/include -D__HIP_PLATFORM_AMD__ $(HOST)/gather.cpp \
	  -L$(ROCM)/lib -lrccl -lamdhip64 -Wl,-rpath,$(ROCM)/lib -o $@

host: $(PKG)/libvolrend_host.a
$(PKG)/libvolrend_host.a: $(HOST_OBJ)
	ar rcs $@ $(HOST_OBJ)

cli: $(PKG)/bin/volrend_headless
$(PKG)/bin/volrend_headless: $(HOST)/main_headless.cpp $(PKG)/libvolrend_host.a $(HOST_HDR) lib gather
	mkdir -p $(PKG)/bin
	$(CXX) -O2 -std=c++17 -Iinclude -I$(ROCM)/include -D__HIP_PLATFORM_AMD__ \
	  $(HOST)/main_headless.cpp $(PKG)/libvolrend_host.a -L$(PKG) -lvolrend_hip -lvolrend_gather \
	  -L$(ROCM)/lib -lamdhip64 -lz -pthread -Wl,-rpath,'$$ORIGIN/..' -Wl,-rpath,$(ROCM)/lib -o $@

oracle:
	$(MAKE) -C oracle
	if [ -d /root/reference ]; then $(MAKE) -C oracle ref; fi

clean:
	rm -f $(HOST_OBJ) $(PKG)/libvolrend_host.a $(PKG)/libvolrend_gather.so $(PKG)/bin/volrend_headless

.PHONY: all lib host gather cli oracle clean
