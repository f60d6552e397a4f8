# Convenience targets; `python -c "import __graft_entry__ as g; g.build()"` does the same and is what the driver runs.
# The library is one host translation unit (oa_icp.hip) + one unit per family of heavy kernel templates (csrc/oa_families.hpp),
# compiled in parallel (make -j8): ~55 s from scratch, ~12 s after an edit that does not touch the brute-force kernels.
HIPCC  ?= /opt/rocm/bin/hipcc
CSRC   := object_alignment_amd/csrc
OBJ    := build/obj
LIB    := object_alignment_amd/liboa_icp.so
LIBEXP := object_alignment_amd/liboa_icp_exp.so
FLAGS  := --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize -fPIC -fvisibility=hidden -pthread -Wall
LINK   := --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -pthread
FAMS   := brute brute_b brute_big brute_big_b grid tri tri_acc bvh affine
FAMOBJ := $(FAMS:%=$(OBJ)/oa_fam_%.o)
HDRS   := $(wildcard $(CSRC)/*.hpp) include/oa_icp.h

all: $(LIB) $(LIBEXP) oracle

$(OBJ)/oa_fam_%.o: $(CSRC)/oa_fam_%.hip $(HDRS)
	@mkdir -p $(OBJ)
	$(HIPCC) $(FLAGS) -c $< -o $@
EXPOBJ := $(OBJ)/oa_fam_exp.o $(OBJ)/oa_fam_exp_r8.o $(OBJ)/oa_fam_exp_r8_big.o
$(EXPOBJ): $(OBJ)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJ)
	$(HIPCC) $(FLAGS) -DOA_EXPERIMENTS -c $< -o $@
$(OBJ)/oa_icp.o: $(CSRC)/oa_icp.hip $(HDRS)
	@mkdir -p $(OBJ)
	$(HIPCC) $(FLAGS) -c $< -o $@
$(OBJ)/oa_icp_exp.o: $(CSRC)/oa_icp.hip $(HDRS)
	@mkdir -p $(OBJ)
	$(HIPCC) $(FLAGS) -DOA_EXPERIMENTS -c $< -o $@

$(LIB): $(OBJ)/oa_icp.o $(FAMOBJ)
	$(HIPCC) $(LINK) -o $@ $^
# the same + the experiments (A/B predecessors, instrumented launches): what tests/test_gpu_*experiment*, *tri_ring*, *sorted_kernel* load
$(LIBEXP): $(OBJ)/oa_icp_exp.o $(EXPOBJ) $(FAMOBJ)
	$(HIPCC) $(LINK) -o $@ $^

oracle:
	$(MAKE) -C oracle

test-cpu: all
	python -m pytest tests -x -q -m "not gpu"

test-gpu: all
	python -m pytest tests -x -q -m gpu

clean:
	rm -rf $(OBJ) $(LIB) $(LIBEXP) oracle/*.so tools/*.exe

.PHONY: all oracle test-cpu test-gpu clean
