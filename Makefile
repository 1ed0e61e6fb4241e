# Build the MI355X (gfx950) C-ABI library and the CPU oracle.
#   make            -> torchregister_amd/lib/libtrx.so  (hipcc cross-compiles without a GPU)
#   make oracle     -> oracle/liboracle.so              (test infrastructure only)
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := torchregister_amd/csrc
SRCS    := $(CSRC)/api.hip $(CSRC)/affine.hip $(CSRC)/flow.hip $(CSRC)/lncc.hip $(CSRC)/kde.hip $(CSRC)/peer.hip
HDRS    := $(CSRC)/trx_common.h $(CSRC)/trx_dev.h $(CSRC)/affine_zstream.h $(CSRC)/affine_eft.h include/trx.h
OBJS    := $(SRCS:$(CSRC)/%.hip=build/%.o)
LIB     := torchregister_amd/lib/libtrx.so
# -fno-slp-vectorize: on gfx950 v_pk_*_f32 is no faster than two scalar VALU ops, and the SLP
# vectoriser's register pairing + v_mov shuffles cost ~50 VGPRs in the tile kernel (occupancy 4 -> 2 waves/SIMD)
HIPFLAGS ?= --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize -Wall -Wno-unused-function -Wno-unused-but-set-variable

all: $(LIB)

build/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p torchregister_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf build $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
