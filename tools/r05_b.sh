#!/bin/bash
# round 5, call B: parity of the changed kernels on the device, then one probe process per variant. Everything under a timeout.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05b; mkdir -p $O
V=$GRAFT_REPO_ROOT/build/variants
timeout 60 build/lone_wave > $O/lone_wave.json 2> $O/lone_wave.err; echo "lone_wave rc=$?" >> $O/rc.txt
timeout 300 python -u -m pytest tests/test_device_stages.py tests/test_pipeline.py -v -x -m gpu -k "not config5 and not 1080p and not large_transforms" > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/rc.txt
probe() { name=$1; shift; ( timeout 150 env "$@" python tools/r05_probe.py 256 16 4 ) >> $O/probes.jsonl 2>> $O/$name.err; echo "$name rc=$?" >> $O/rc.txt; }
probe base A=1
probe lf_lanes J40HIP_LF_KERNEL=lanes PROBE_ONLY=lf_alone
probe lf_lanes_dev J40HIP_LF_KERNEL=lanes PROBE_ONLY=device
probe rows24 J40HIP_LF_ROWS_LDS_KB=24 PROBE_ONLY=lf_alone
probe ev0 J40HIP_LIB=$V/libj40hip_ev0.so
probe k2w8 J40HIP_LIB=$V/libj40hip_k2w8.so PROBE_ONLY=alone
pmc() { name=$1; shift; ( cd /tmp && timeout 200 env "$@" PROBE_ONLY=alone rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_$name --output-format csv -- python $GRAFT_REPO_ROOT/tools/r05_probe.py 256 16 > $O/pmc_$name.log 2>&1 ); echo "pmc_$name rc=$?" >> $O/rc.txt
	python tools/pmc_summary.py /tmp/pmc_$name $O/pmc_write_$name.txt > /dev/null 2>&1; rm -rf /tmp/pmc_$name; }
pmc base A=1
pmc ev0 J40HIP_LIB=$V/libj40hip_ev0.so
cat $O/rc.txt; cat $O/lone_wave.json; grep -c PASSED $O/tests.txt; grep -E 'FAILED|Error|passed|failed' $O/tests.txt | tail -n 6; cat $O/probes.jsonl; grep -A2 "k_hf_lanes" $O/pmc_write_base.txt $O/pmc_write_ev0.txt
