# Builds the product library (build/libj40hip.so: host parser + HIP kernels for gfx950 + public API),
# the synthetic stream generator (build/jxlsynth) and the checkers under oracle/.
ROCM ?= /opt/rocm
HIPCC ?= $(ROCM)/bin/hipcc
CXX ?= g++
ARCH ?= gfx950
CXXFLAGS = -std=c++17 -O2 -fPIC -Wall -Wextra -ffp-contract=off -fvisibility=hidden
# EVENT_RING=8 (or 4): k_hf_lanes' coefficient events leave through per-lane rings in LDS as aligned 32- (16-) byte stores
# (device/plan.h: J40_LANE_EV_FLUSH; default 0 = a 4-byte store per event). The CPU build of the device functions exists in both
# forms: libhostsim.so with the product's setting, libhostsim_ring8.so with the rings (tests/test_hostsim.py runs through both).
EVENT_RING ?= 0
CXXFLAGS += -DJ40_LANE_EV_FLUSH=$(EVENT_RING)
HIPFLAGS = --offload-arch=$(ARCH) -std=c++17 -O3 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden -Wall -DJ40_LANE_EV_FLUSH=$(EVENT_RING) $(EXTRA_HIPFLAGS)
SRC = j40_amd/csrc
HOST_OBJS = build/obj/plan_build.o build/obj/plan_front.o build/obj/entropy.o build/obj/modular.o build/obj/tables.o build/obj/frame.o build/obj/capi_host.o build/obj/api.o
DEV_OBJS = build/obj/kernels.o build/obj/modular_kernels.o build/obj/runtime.o build/obj/pipeline.o build/obj/lf_tail_kernels.o build/obj/modular_coop.o build/obj/modular_quad.o build/obj/modular_split.o build/obj/lf_decode.o build/obj/plan_kernels.o build/obj/async.o build/obj/hostcopy.o

.PHONY: all lib tools oracle hostsim clean
all: lib tools hostsim oracle
lib: build/libj40hip.so
tools: build/jxlsynth
oracle:
	$(MAKE) -C oracle all

build/obj/%.o: $(SRC)/%.cpp $(wildcard $(SRC)/*.hpp) $(wildcard $(SRC)/device/*.h) include/j40hip.h include/j40.h
	@mkdir -p build/obj
	$(CXX) $(CXXFLAGS) -c $< -o $@

build/obj/%.o: $(SRC)/device/%.hip $(wildcard $(SRC)/device/*.h) $(wildcard $(SRC)/*.hpp) include/j40hip.h
	@mkdir -p build/obj
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# lf_decode.hip: k_lf_rows steps two LfGroup sections per lane side by side; the two sections' instructions are independent and the
# scheduler is asked to interleave them rather than to keep register pressure low (a wavefront alone on its SIMD, 512 registers its own)
build/obj/lf_decode.o: EXTRA_HIPFLAGS += -mllvm -amdgpu-sched-strategy=max-ilp

build/libj40hip.so: $(HOST_OBJS) $(DEV_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -o $@ $^ -lpthread -lhsa-runtime64

hostsim: build/libhostsim.so build/libhostsim_ring8.so build/liboracle_driver.so build/api_threads
# test-only glue: parses a stream with the product's host parser, takes the plan view and hands it to
# the CPU oracle (oracle/libj40oracle.so)
build/liboracle_driver.so: tests/oracle_driver.c build/libj40hip.so oracle/hotpath_oracle.c include/j40hip.h
	$(MAKE) -C oracle restatement
	gcc -O2 -fPIC -shared -Wall -o $@ tests/oracle_driver.c -Lbuild -Loracle -lj40hip -lj40oracle -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,'$$ORIGIN/../oracle'

# test-only: many threads running the reference's public API sequence (dj40.c's) against the product library
build/api_threads: tests/api_threads.c include/j40.h build/libj40hip.so
	gcc -O2 -Wall -Wextra -pthread -Iinclude -o $@ tests/api_threads.c -Lbuild -lj40hip -Wl,-rpath,'$$ORIGIN'

# device functions compiled for the CPU, test infrastructure only (tests/hostsim)
HOSTSIM_SRC = tests/hostsim/hostsim.cpp $(SRC)/plan_build.cpp $(SRC)/plan_front.cpp $(SRC)/entropy.cpp $(SRC)/modular.cpp $(SRC)/tables.cpp $(SRC)/frame.cpp
HOSTSIM_FLAGS = -std=c++17 -O2 -fPIC -shared -ffp-contract=off -Wall -Wextra -Wno-unused-function -Wno-unknown-pragmas
build/libhostsim.so: $(HOSTSIM_SRC) $(wildcard $(SRC)/device/*.h) $(wildcard $(SRC)/*.hpp)
	@mkdir -p build
	$(CXX) $(HOSTSIM_FLAGS) -DJ40_LANE_EV_FLUSH=$(EVENT_RING) -o $@ $(HOSTSIM_SRC) -lpthread
build/libhostsim_ring8.so: $(HOSTSIM_SRC) $(wildcard $(SRC)/device/*.h) $(wildcard $(SRC)/*.hpp)
	@mkdir -p build
	$(CXX) $(HOSTSIM_FLAGS) -DJ40_LANE_EV_FLUSH=8 -o $@ $(HOSTSIM_SRC) -lpthread

build/jxlsynth: tools/jxlsynth.cpp $(wildcard tools/*.hpp) $(SRC)/tables.cpp $(SRC)/device/special8_dev.h $(SRC)/device/idct_dev.h
	@mkdir -p build
	$(CXX) -O2 -std=c++17 -ffp-contract=off -Wall -Wextra -Wno-unused-function -Wno-unknown-pragmas -o $@ $< $(SRC)/tables.cpp

clean:
	rm -rf build
	$(MAKE) -C oracle clean
