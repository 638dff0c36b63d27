#!/bin/bash
# round 4: A/B of variant libraries with the stages alone on the device (one batch in flight); the LfGroup lane decoder's tables in LDS
# with one frame per wavefront, on the device_output clock. Writes gpurun_out/r04AB2/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04AB2; mkdir -p $O
V=$GRAFT_REPO_ROOT/build/variants
run() { name=$1; shift; ( "$@" ) >> $O/$name.json 2>> $O/$name.err; echo "$name rc=$?" >> $O/rc.txt; }
for i in 1 2 3; do
run alone_base timeout 60 python tools/stages_alone_probe.py 256 4 8
J40HIP_LIB=$V/libj40hip_colour2.so run alone_colour2 timeout 60 python tools/stages_alone_probe.py 256 4 8
done
for i in 1 2; do
run dev_base timeout 90 python tools/device_output_probe.py 256 12 device 2 8
J40HIP_LF_ALIAS_LDS=1 J40HIP_LF_LDS_KB=30 run dev_lf_alias_lds30 timeout 90 python tools/device_output_probe.py 256 12 device 2 8
done
cat $O/rc.txt; cat $O/*.json
