#!/bin/bash
# round 4: A/B measurements of experimental variants in one call (tools/build_variant.sh). Writes gpurun_out/r04AB/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04AB; mkdir -p $O
V=$GRAFT_REPO_ROOT/build/variants
run() { name=$1; shift; ( "$@" ) >> $O/$name.json 2>> $O/$name.err; echo "$name rc=$?" >> $O/rc.txt; }
run large_base timeout 60 python tools/large_probe.py 64 3 16
J40HIP_LIB=$V/libj40hip_large512.so run large_512 timeout 60 python tools/large_probe.py 64 3 16
for i in 1 2; do
run dev_base timeout 90 python tools/device_output_probe.py 256 6 device 2 8
J40HIP_LIB=$V/libj40hip_colour2.so run dev_colour2 timeout 90 python tools/device_output_probe.py 256 6 device 2 8
done
J40HIP_LF_ALIAS_LDS=1 J40HIP_LF_LDS_KB=30 run dev_lf_alias_lds30 timeout 90 python tools/device_output_probe.py 256 6 device 2 8
J40HIP_LF_ALIAS_LDS=1 run dev_lf_alias_lds56 timeout 90 python tools/device_output_probe.py 256 6 device 2 8
J40HIP_LIB=$V/libj40hip_colour2.so timeout 120 python -m pytest tests/test_gpu_parity.py tests/test_pipeline.py -q -x -m gpu -k "srgb or batches_give or all_transforms" > $O/test_colour2.txt 2>&1; echo "colour2 tests rc=$?" >> $O/rc.txt
cat $O/rc.txt; cat $O/*.json; tail -n 3 $O/test_colour2.txt
