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

$(B)/libgroot_hip.so: $(wildcard groot_amd/csrc/hip/*) $(wildcard groot_amd/csrc/common/*.hpp) $(wildcard include/*.h)
	@mkdir -p $(B)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-pass-failed -Iinclude -Igroot_amd/csrc/hip -o $@ groot_amd/csrc/hip/groot_hip.hip

$(B)/groot-hip: groot_amd/csrc/cli/groot_hip_main.cpp $(B)/libgroot_host.so $(B)/libgroot_hip.so
	g++ -O2 -std=c++17 -Wall -Wextra -Iinclude -o $@ $< -L$(B) -lgroot_hip -lgroot_host -lpthread \
	    '-Wl,-rpath,$$ORIGIN' -Wl,-rpath-link,$(B) -Wl,-rpath-link,/opt/rocm/lib

oracle:
	$(MAKE) -C oracle -s

clean:
	rm -f $(B)/libgroot_host.so $(B)/libgroot_hip.so $(B)/groot-hip

.PHONY: all oracle clean
