#!/bin/bash
# round 5, call D: two LfGroup sections per lane (k_lf_rows<true>) against one, the event ring's scalar tick, config 5 again.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05d; mkdir -p $O
V=$GRAFT_REPO_ROOT/build/variants
timeout 300 python -u -m pytest tests/test_device_stages.py tests/test_pipeline.py tests/test_api_threads.py -v -x -m gpu -k "not config5 and not large_transforms and not queued and not 64_threads and not mix_of_images" > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/rc.txt
probe() { name=$1; shift; ( timeout 150 env "$@" python tools/r05_probe.py 256 16 4 ) >> $O/probes.jsonl 2>> $O/$name.err; echo "$name rc=$?" >> $O/rc.txt; }
probe base A=1
probe rows1 J40HIP_LF_KERNEL=rows1 PROBE_ONLY=lf_alone
probe rows1_dev J40HIP_LF_KERNEL=rows1 PROBE_ONLY=device
probe ev0 J40HIP_LIB=$V/libj40hip_ev0.so PROBE_ONLY=alone
probe rows60 J40HIP_LF_ROWS_LDS_KB=60 PROBE_ONLY=lf_alone
probe rows60_dev J40HIP_LF_ROWS_LDS_KB=60 PROBE_ONLY=device
probe steps12 PROBE_ONLY=device PROBE_STEPS=12
run() { name=$1; shift; ( "$@" ) >> $O/$name.json 2>> $O/$name.err; echo "$name rc=$?" >> $O/rc.txt; }
run c5_host timeout 100 python tools/config5_probe.py 512 2 host
run c5_device timeout 100 python tools/config5_probe.py 512 2 device 4
cat $O/rc.txt; grep -c PASSED $O/tests.txt; grep -E 'FAILED|Error|passed|failed' $O/tests.txt | tail -n 6; cat $O/probes.jsonl; cat $O/c5_*.json
