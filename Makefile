# Top-level build: everything is built IN-TREE so the artefacts travel with the repo snapshot.
#
#   cuda-raytracing-optimized_amd/librt_mi355x.so   HIP renderer behind the reference's C-ABI (gfx950 only)
#   cuda-raytracing-optimized_amd/librt_host.so     host-side scene / BVH / PPM / .ref harness (plain C++, no HIP)
#   oracle/liboracle.so, oracle/_ref/libref.so      CPU checker (test infrastructure; see oracle/Makefile)

PKG      := cuda-raytracing-optimized_amd
HIPCC    ?= hipcc
CXX      ?= g++
ARCH     ?= gfx950
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function
CSRC     := $(PKG)/csrc
HOST     := $(PKG)/host
OBJ      := build/obj

KERNEL_HDRS := $(CSRC)/rt_device.h $(CSRC)/rt_div64.h $(CSRC)/rt_glibc_sincosf.h $(CSRC)/rt_glibc_powf.h $(CSRC)/rt_params.h include/rt_types.h

all: $(PKG)/librt_mi355x.so $(PKG)/librt_host.so
	$(MAKE) -C oracle

$(OBJ):
	@mkdir -p $(OBJ)

# The two floating-point builds of each kernel TU: PARITY never contracts a*b+c into an FMA.
# -fno-slp-vectorize -fno-vectorize (PARITY kernels): the vectorisers pair the un-fused mul/add of two boxes / two spheres into v_pk_mul_f32 /
# v_pk_add_f32.  On gfx950 a packed mul/add issues in 4.56 cycles against 2 x 2.3 for the scalar pair (tools/valu_microbench.hip), so
# it gains nothing, and the register shuffles (v_mov) that feed it cost 20 % of the box test.  FAST keeps it: v_pk_fma_f32 does pay.
$(OBJ)/spheres_parity.o: $(CSRC)/rt_kernels_spheres.hip $(KERNEL_HDRS) | $(OBJ)
	$(HIPCC) $(HIPFLAGS) -DRT_MODE_PARITY -ffp-contract=off -fno-slp-vectorize -fno-vectorize -c $< -o $@
$(OBJ)/spheres_fast.o: $(CSRC)/rt_kernels_spheres.hip $(KERNEL_HDRS) | $(OBJ)
	$(HIPCC) $(HIPFLAGS) -DRT_MODE_FAST -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -fno-vectorize -c $< -o $@
$(OBJ)/mesh_parity.o: $(CSRC)/rt_kernels_mesh.hip $(KERNEL_HDRS) | $(OBJ)
	$(HIPCC) $(HIPFLAGS) -DRT_MODE_PARITY -ffp-contract=off -fno-slp-vectorize -fno-vectorize -c $< -o $@
$(OBJ)/mesh_fast.o: $(CSRC)/rt_kernels_mesh.hip $(KERNEL_HDRS) | $(OBJ)
	$(HIPCC) $(HIPFLAGS) -DRT_MODE_FAST -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -c $< -o $@
$(OBJ)/probe_parity.o: $(CSRC)/rt_probe.hip $(KERNEL_HDRS) include/rt_probe.h | $(OBJ)
	$(HIPCC) $(HIPFLAGS) -DRT_MODE_PARITY -ffp-contract=off -c $< -o $@
$(OBJ)/probe_fast.o: $(CSRC)/rt_probe.hip $(KERNEL_HDRS) include/rt_probe.h | $(OBJ)
	$(HIPCC) $(HIPFLAGS) -DRT_MODE_FAST -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -c $< -o $@
$(OBJ)/renderer.o: $(CSRC)/rt_renderer.hip $(CSRC)/rt_params.h include/rt_api.h include/rt_types.h | $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

RT_OBJS := $(OBJ)/renderer.o $(OBJ)/probe_parity.o $(OBJ)/probe_fast.o $(OBJ)/spheres_parity.o $(OBJ)/spheres_fast.o $(OBJ)/mesh_parity.o $(OBJ)/mesh_fast.o

$(PKG)/librt_mi355x.so: $(RT_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(RT_OBJS) -o $@

$(PKG)/librt_host.so: $(HOST)/rt_scenes.cpp $(HOST)/rt_bvh.cpp $(HOST)/rt_harness.cpp include/rt_host.h include/rt_types.h
	$(CXX) -O2 -ffp-contract=off -std=c++14 -Wall -fPIC -shared $(HOST)/rt_scenes.cpp $(HOST)/rt_bvh.cpp $(HOST)/rt_harness.cpp -o $@

oracle: $(PKG)/librt_mi355x.so $(PKG)/librt_host.so
	$(MAKE) -C oracle

clean:
	rm -rf build/obj $(PKG)/librt_mi355x.so $(PKG)/librt_host.so
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
