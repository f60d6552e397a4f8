# Convenience targets; `python -c "import __graft_entry__ as g; g.build()"` does the same and is what the driver runs.
HIPCC  ?= /opt/rocm/bin/hipcc
CSRC   := object_alignment_amd/csrc
LIB    := object_alignment_amd/liboa_icp.so
FLAGS  := --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize -fPIC -shared -fvisibility=hidden -pthread -Wall

all: $(LIB) oracle

$(LIB): $(CSRC)/oa_icp.hip $(wildcard $(CSRC)/*.hpp) include/oa_icp.h
	$(HIPCC) $(FLAGS) -o $@ $(CSRC)/oa_icp.hip

oracle:
	$(MAKE) -C oracle

test-cpu: all
	python -m pytest tests -x -q -m "not gpu"

test-gpu: all
	python -m pytest tests -x -q -m gpu

clean:
	rm -f $(LIB) oracle/*.so tools/*.exe

.PHONY: all oracle test-cpu test-gpu clean
