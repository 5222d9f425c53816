# Builds librtxn.so (HIP, gfx950 only) and the CPU oracle used by the tests.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := rtx_nerf_amd/csrc
SRCS := $(wildcard $(CSRC)/*.hip)
OBJS := $(patsubst $(CSRC)/%.hip,build/%.o,$(SRCS))
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -ffp-contract=off -Iinclude -I$(CSRC) -Wall -Wno-unused-function

all: rtx_nerf_amd/librtxn.so oracle

build/%.o: $(CSRC)/%.hip $(CSRC)/common.h include/rtxn.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

rtx_nerf_amd/librtxn.so: $(OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(OBJS)

oracle:
	$(MAKE) -C oracle -s

clean:
	rm -rf build rtx_nerf_amd/librtxn.so
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
