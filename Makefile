# Builds the C-ABI kernel library for MI355X (gfx950) in-tree, and the C parts of the oracle.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := noise_amd/csrc
SRCS    := $(wildcard $(CSRC)/*.hip)
OBJS    := $(SRCS:.hip=.o)
LIB     := noise_amd/libnoise_hip.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -fvisibility=hidden -Wall -Wno-unused-function -Wno-pass-failed \
            -ffp-contract=off -fno-fast-math -Iinclude

all: $(LIB)

$(CSRC)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/noise_hip.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -f $(OBJS) $(LIB)

.PHONY: all clean
