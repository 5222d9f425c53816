# Builds librtxn.so (HIP, gfx950 only) and the CPU oracle used by the tests.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := rtx_nerf_amd/csrc
SRCS := $(wildcard $(CSRC)/*.hip)
OBJS := $(patsubst $(CSRC)/%.hip,build/%.o,$(SRCS)) build/loader.o
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -ffp-contract=off -Iinclude -I$(CSRC) -Wall -Wno-unused-function

all: rtx_nerf_amd/librtxn.so oracle examples/render_host examples/render_host_mgpu examples/train_host examples/train_host_mgpu check-isa

# train.hip: -amdgpu-mfma-vgpr-form.  Its one kernel above 256 registers (mlp_bwd_fused64_kernel, one wave per SIMD) keeps the
# weight-gradient accumulators in AGPRs by hand (asm "+a"); left to its heuristic, hipcc put the destination of EVERY MFMA of
# that kernel in AGPRs and copied each chain accumulator back for the fp16 conversion (816 v_accvgpr_read per tile).
build/train.o: EXTRA := -mllvm -amdgpu-mfma-vgpr-form
build/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/mlp_internal.h $(CSRC)/hashgrid_internal.h include/rtxn.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) $(EXTRA) -c $< -o $@

# ISA text of the files that hold asm MFMAs, and the static check that no compiler-generated instruction reads an asm MFMA's
# result inside its wait states (tools/check_asm_mfma_reads.py; the round-3 hoisted-conversion bug)
build/train.s: EXTRA := -mllvm -amdgpu-mfma-vgpr-form
build/%.s: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/mlp_internal.h $(CSRC)/hashgrid_internal.h include/rtxn.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) $(EXTRA) -S --cuda-device-only $< -o $@
check-isa: build/train.s build/mlp.s build/hashmlp.s
	python3 tools/check_asm_mfma_reads.py $^
	python3 tools/check_asm_sgpr_hazard.py $^

build/loader.o: $(CSRC)/loader.cpp $(CSRC)/common.h include/rtxn.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

rtx_nerf_amd/librtxn.so: $(OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(OBJS) -lz

oracle:
	$(MAKE) -C oracle -s

# C++ host over the drop-in headers (reference-style call sites); links librtxn.so by rpath
DROPIN := -Iinclude -Iinclude/rtxn_dropin -Iinclude/rtxn_dropin/sampler -Iinclude/rtxn_dropin/vol_render -Iinclude/rtxn_dropin/loader
examples/render_host: examples/render_host.cpp rtx_nerf_amd/librtxn.so $(wildcard include/rtxn_dropin/*/*.h include/rtxn_dropin/rtx/include/*.h)
	$(HIPCC) -O2 -std=c++17 --offload-arch=$(ARCH) $(DROPIN) $< -o $@ -Lrtx_nerf_amd -lrtxn -Wl,-rpath,'$$ORIGIN/../rtx_nerf_amd'

examples/render_host_mgpu: examples/render_host_mgpu.cpp rtx_nerf_amd/librtxn.so include/rtxn.h
	$(HIPCC) -O2 -std=c++17 --offload-arch=$(ARCH) -Iinclude $< -o $@ -Lrtx_nerf_amd -lrtxn -Wl,-rpath,'$$ORIGIN/../rtx_nerf_amd'

examples/train_host: examples/train_host.cpp rtx_nerf_amd/librtxn.so include/rtxn.h
	$(HIPCC) -O2 -std=c++17 --offload-arch=$(ARCH) -Iinclude $< -o $@ -Lrtx_nerf_amd -lrtxn -Wl,-rpath,'$$ORIGIN/../rtx_nerf_amd'

examples/train_host_mgpu: examples/train_host_mgpu.cpp rtx_nerf_amd/librtxn.so include/rtxn.h
	$(HIPCC) -O2 -std=c++17 --offload-arch=$(ARCH) -Iinclude $< -o $@ -Lrtx_nerf_amd -lrtxn -Wl,-rpath,'$$ORIGIN/../rtx_nerf_amd'

clean:
	rm -rf build rtx_nerf_amd/librtxn.so examples/render_host examples/render_host_mgpu examples/train_host examples/train_host_mgpu
	$(MAKE) -C oracle clean

.PHONY: all oracle clean check-isa
