#!/bin/bash
# round 5, call A: lone-wavefront instruction timings; the row-window LfGroup decoder (k_lf_rows) against the older k_lf_lanes --
# parity on the device, then kernel traces of the pipeline with the pixels left in HBM. Writes gpurun_out/r05a/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05a; mkdir -p $O
build/lone_wave > $O/lone_wave.json 2> $O/lone_wave.err; echo "lone_wave rc=$?" >> $O/rc.txt
timeout 900 python -m pytest tests/test_device_stages.py tests/test_pipeline.py -q -x -m gpu > $O/tests_rows.txt 2>&1; echo "tests rc=$?" >> $O/rc.txt
kt() { name=$1; shift; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$name -- "$@" > $O/kt_$name.log 2>&1 ); echo "kt_$name rc=$?" >> $O/rc.txt
	python tools/prof_summary.py /tmp/kt_$name $O/kernel_stats_$name.txt > /dev/null 2>&1; rm -rf /tmp/kt_$name; grep -h "mpixels_per_s" $O/kt_$name.log >> $O/probes.jsonl; }
P="python $GRAFT_REPO_ROOT/tools/device_output_probe.py 256 4 device 2 16"
J40HIP_LF_KERNEL=lanes kt lanes $P
kt rows48 $P
J40HIP_LF_ROWS_LDS_KB=24 kt rows24 $P
J40HIP_LF_ROWS_LDS_KB=60 kt rows60 $P
run() { name=$1; shift; ( "$@" ) >> $O/$name.json 2>> $O/$name.err; echo "$name rc=$?" >> $O/rc.txt; }
run c5_host timeout 120 python tools/config5_probe.py 256 4 host
run c5_device timeout 120 python tools/config5_probe.py 256 4 device 4
J40HIP_LF_KERNEL=lanes run c5_device_lanes timeout 120 python tools/config5_probe.py 256 4 device 4
cat $O/rc.txt; cat $O/lone_wave.json; tail -n 3 $O/tests_rows.txt; cat $O/probes.jsonl; cat $O/c5_*.json
for v in lanes rows48 rows24 rows60; do grep -h "k_lf_\|k_hf_lanes\|kernel " $O/kernel_stats_$v.txt | cut -c1-60,108-190; done
