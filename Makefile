# Builds libccqp_hip.so (the C-ABI HIP engine, gfx950 only) in-tree:
#   ccqppy_amd/libccqp_hip.so   -- git-ignored, travels to the GPU box with the snapshot.
# hipcc cross-compiles without a GPU.  No rpath to /opt/rocm is baked in: inside Python the
# library binds to the HIP runtime torch already loaded (one runtime per process).
HIPCC    ?= hipcc
ARCH     ?= gfx950
ROCM_LIB ?= /opt/rocm/lib
CSRC     := ccqppy_amd/csrc
SRCS     := $(wildcard $(CSRC)/*.hip)
OBJS     := $(patsubst $(CSRC)/%.hip,build/%.o,$(SRCS))
LIB      := ccqppy_amd/libccqp_hip.so
# -ffp-contract=off: elementwise kernels must round once per operation like numpy; the matvec
# and the reductions call fma() explicitly.
CXXFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -I$(CSRC) \
            -Wall -Wno-unused-function

all: $(LIB)

build/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/ccqp_hip.h
	@mkdir -p build
	$(HIPCC) $(CXXFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -no-hip-rt -o $@ $(OBJS) \
	    -L$(ROCM_LIB) -lamdhip64 -Wl,--no-undefined

clean:
	rm -rf build $(LIB)

.PHONY: all clean
