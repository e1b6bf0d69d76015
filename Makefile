# Top-level build: the product library (HIP, gfx950) and the parity checkers (oracle/).
#
#   make            -> gr-fosphor_amd/libfosphor_amd.so  (hipcc cross-compiles without a GPU)
#   make oracle     -> oracle/libfosphor_oracle.so, oracle/pm_check   (gcc)
#   make ref        -> oracle/_ref/libfosphor_ref.so (needs /root/reference)
#
# -ffp-contract=off is part of the numerical contract: the FFT must round
# exactly where the reference's expressions round.

HIPCC   ?= hipcc
ARCH    ?= gfx950
CSRC    := gr-fosphor_amd/csrc
LIB     := gr-fosphor_amd/libfosphor_amd.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -ffp-contract=off -std=c++17 -fPIC -pthread -Wall -Wno-unused-function

SRCS := $(CSRC)/fosphor_kernels.hip $(CSRC)/fosphor_cmap.hip $(CSRC)/fosphor_api.cpp $(CSRC)/fosphor_render.cpp $(CSRC)/fosphor_sink.cpp $(CSRC)/fosphor_exchange.cpp
HDRS := $(CSRC)/fosphor_internal.h include/fosphor.h include/fosphor_amd.h include/fosphor_amd_sink.h include/fosphor_amd_cmap.h include/fosphor_amd_axis.h include/fosphor_portable_math.h

all: $(LIB)

$(LIB): $(SRCS) $(HDRS) Makefile
	$(HIPCC) $(HIPFLAGS) -x hip -shared -o $@ $(SRCS) -ldl

oracle:
	$(MAKE) -C oracle all

ref:
	$(MAKE) -C oracle ref

clean:
	rm -f $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle ref clean
