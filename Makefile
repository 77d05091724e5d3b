# Build libddx_hip.so (gfx950 only).  `python -c "import __graft_entry__ as g; g.build()"` calls this.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := dualdiffusion_amd/csrc
OBJ   := build/obj
LIB   := dualdiffusion_amd/lib/libddx_hip.so
SRCS  := $(wildcard $(CSRC)/*.hip)
OBJS  := $(patsubst $(CSRC)/%.hip,$(OBJ)/%.o,$(SRCS))
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=fast

all: $(LIB)

$(OBJ)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.hpp) include/ddx_hip.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p $(dir $(LIB))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

# one-off hardware probes quoted in DESIGN.md (run on the GPU box)
probes: tools/probe/bufload_lds_probe tools/probe/ds_read_tr_probe tools/probe/launch_floor_probe tools/probe/stage_bw_probe tools/probe/grid_barrier_probe tools/probe/store_pattern_probe tools/probe/lds_stride_probe
tools/probe/%: tools/probe/%.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 $< -o $@

clean:
	rm -rf build $(LIB)

.PHONY: all clean probes
