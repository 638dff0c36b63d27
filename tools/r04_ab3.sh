#!/bin/bash
# round 4: A/B of the event scatter with several events per lane in flight (J40_SCATTER_U), stages alone on the device. Writes gpurun_out/r04AB3/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04AB3; mkdir -p $O
V=$GRAFT_REPO_ROOT/build/variants
run() { name=$1; shift; ( "$@" ) >> $O/$name.json 2>> $O/$name.err; echo "$name rc=$?" >> $O/rc.txt; }
for i in 1 2; do
run alone_base timeout 60 python tools/stages_alone_probe.py 256 3 8
J40HIP_LIB=$V/libj40hip_scatter4.so run alone_scatter4 timeout 60 python tools/stages_alone_probe.py 256 3 8
J40HIP_LIB=$V/libj40hip_scatter2.so run alone_scatter2 timeout 60 python tools/stages_alone_probe.py 256 3 8
done
cat $O/rc.txt; cat $O/*.json
