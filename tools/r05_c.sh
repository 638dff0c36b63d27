#!/bin/bash
# round 5, call C: the straight-line LfGroup step on the device (parity, then timings), the event ring at 0 / 4 / 8 events per store,
# config 5 with the LfGroup streams on either side, the new tests. Everything under a timeout.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05c; mkdir -p $O
V=$GRAFT_REPO_ROOT/build/variants
timeout 240 python -u -m pytest tests/test_device_stages.py tests/test_pipeline.py -v -x -m gpu -k "not config5 and not 1080p and not large_transforms and not queued" > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/rc.txt
probe() { name=$1; shift; ( timeout 150 env "$@" python tools/r05_probe.py 256 16 4 ) >> $O/probes.jsonl 2>> $O/$name.err; echo "$name rc=$?" >> $O/rc.txt; }
probe base A=1
probe ev4 J40HIP_LIB=$V/libj40hip_ev4.so
probe ev0 J40HIP_LIB=$V/libj40hip_ev0.so
probe rows24 J40HIP_LF_ROWS_LDS_KB=24 PROBE_ONLY=device
probe ev4_rows24 J40HIP_LIB=$V/libj40hip_ev4.so J40HIP_LF_ROWS_LDS_KB=24 PROBE_ONLY=device
run() { name=$1; shift; ( "$@" ) >> $O/$name.json 2>> $O/$name.err; echo "$name rc=$?" >> $O/rc.txt; }
run c5_host timeout 100 python tools/config5_probe.py 512 2 host
run c5_device timeout 100 python tools/config5_probe.py 512 2 device 4
( cd /tmp && timeout 150 env J40HIP_LIB=$V/libj40hip_ev4.so PROBE_ONLY=alone rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_ev4 --output-format csv -- python $GRAFT_REPO_ROOT/tools/r05_probe.py 256 16 > $O/pmc_ev4.log 2>&1 ); python tools/pmc_summary.py /tmp/pmc_ev4 $O/pmc_write_ev4.txt > /dev/null 2>&1
timeout 240 python -u -m pytest tests/test_bench_ranks.py tests/test_api_threads.py -v -x -m gpu -k "two_ranks or pinned" > $O/tests_new.txt 2>&1; echo "tests_new rc=$?" >> $O/rc.txt
cat $O/rc.txt; grep -c PASSED $O/tests.txt; grep -E 'FAILED|Error|passed|failed' $O/tests.txt | tail -n 6; cat $O/probes.jsonl; cat $O/c5_*.json; grep -A1 "k_hf_lanes" $O/pmc_write_ev4.txt; grep -E 'PASSED|FAILED|Error|passed|failed' $O/tests_new.txt | tail -n 8
