# Same commands as __graft_entry__.build(), for maintainers who build without Python.
#   make            libgroot_host.so, libgroot_hip.so (gfx950), groot-hip
#   make oracle     the CPU checker used by the tests only
HIPCC ?= /opt/rocm/bin/hipcc
B := build
HOST_SRC := $(addprefix groot_amd/csrc/host/,index.cpp gob.cpp graphs.cpp fastq.cpp reads.cpp bam.cpp report.cpp)

all: $(B)/libgroot_host.so $(B)/libgroot_hip.so $(B)/groot-hip

$(B)/libgroot_host.so: $(HOST_SRC) groot_amd/csrc/host/host_common.hpp $(wildcard groot_amd/csrc/common/*.hpp) $(wildcard include/*.h)
	@mkdir -p $(B)
	g++ -O2 -std=c++17 -fPIC -Wall -Wextra -Iinclude -shared -o $@ $(HOST_SRC) -lpthread -lz

# libgroot_hip.so: four translation units (groot_amd/csrc/hip/launch.hpp) compiled side by side (make -j4)
HIP_DIR := groot_amd/csrc/hip
HIP_TU := groot_hip seed_full seed_fast align
HIP_OBJ := $(addprefix $(B)/obj/,$(addsuffix .o,$(HIP_TU)))
HIP_FLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -Iinclude -I$(HIP_DIR)
$(B)/obj/%.o: $(HIP_DIR)/%.hip $(wildcard $(HIP_DIR)/*.hpp) $(wildcard groot_amd/csrc/common/*.hpp) $(wildcard include/*.h)
	@mkdir -p $(B)/obj
	$(HIPCC) $(HIP_FLAGS) -c -o $@ $<
$(B)/libgroot_hip.so: $(HIP_OBJ)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(HIP_OBJ)

$(B)/groot-hip: groot_amd/csrc/cli/groot_hip_main.cpp $(B)/libgroot_host.so $(B)/libgroot_hip.so
	g++ -O2 -std=c++17 -Wall -Wextra -Iinclude -o $@ $< -L$(B) -lgroot_hip -lgroot_host -lpthread \
	    '-Wl,-rpath,$$ORIGIN' -Wl,-rpath-link,$(B) -Wl,-rpath-link,/opt/rocm/lib

oracle:
	$(MAKE) -C oracle -s

clean:
	rm -rf $(B)/obj $(B)/libgroot_host.so $(B)/libgroot_hip.so $(B)/groot-hip

.PHONY: all oracle clean
