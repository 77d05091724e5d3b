# Build libddx_hip.so (gfx950 only) and the C oracle helpers.  `python -c "import __graft_entry__ as g; g.build()"` calls this.
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

clean:
	rm -rf build $(LIB)

.PHONY: all clean
