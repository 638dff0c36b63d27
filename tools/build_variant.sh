#!/bin/bash
# an experimental variant of the product library beside the real one: tools/build_variant.sh NAME kernels.hip [tables.cpp ...] -DFLAG ...
# compiles the named device sources with the extra flags and links them with the other objects of build/obj into
# build/variants/libj40hip_NAME.so (select it with J40HIP_LIB=...). For A/B measurements in one gpurun call; never shipped.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
FLAGS=""
SRCS=()
for a in "$@"; do case "$a" in *.hip|*.cpp) SRCS+=("$a");; *) FLAGS="$FLAGS $a";; esac; done
mkdir -p build/variants build/obj/variant_$NAME
OBJS=""
for o in plan_build plan_front entropy modular tables frame capi_host api kernels modular_kernels runtime pipeline lf_tail_kernels modular_coop modular_quad modular_split lf_decode plan_kernels async hostcopy; do
	use=build/obj/$o.o
	for s in "${SRCS[@]}"; do
		if [ "$s" = "$o.hip" ]; then
			use=build/obj/variant_$NAME/$o.o
			PER=""; if [ "$o" = "lf_decode" ]; then PER="-mllvm -amdgpu-sched-strategy=max-ilp"; fi   # (as the Makefile)
			/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden -Wall -DJ40_LANE_EV_FLUSH=0 $PER $FLAGS -c j40_amd/csrc/device/$o.hip -o $use
		elif [ "$s" = "$o.cpp" ]; then
			use=build/obj/variant_$NAME/$o.o
			g++ -std=c++17 -O2 -fPIC -Wall -Wextra -ffp-contract=off -fvisibility=hidden $FLAGS -c j40_amd/csrc/$o.cpp -o $use
		fi
	done
	OBJS="$OBJS $use"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/variants/libj40hip_$NAME.so $OBJS -lpthread -lhsa-runtime64
echo built build/variants/libj40hip_$NAME.so
