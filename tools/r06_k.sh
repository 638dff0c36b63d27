#!/bin/bash
# round 6, call K: k_lf_rows with fast entries (the hybrid integer worked out per symbol when the tables are staged), deferred checks
# and counted runs -- the device-stage and pipeline tests, the damaged-stream sweep through the pipeline, then the launch alone on
# the device (PROBE_ONLY=lf_alone), the pixels-in-HBM pipeline, and a kernel trace of the launch alone
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06k; mkdir -p $O
( timeout 1500 python -m pytest tests/test_device_stages.py tests/test_pipeline.py tests/test_lf_decoder_glue.py -m gpu -x -q ) > $O/pytest.txt 2>&1; echo "tests rc=$?" >> $O/rc.txt
tail -5 $O/pytest.txt
( timeout 900 python tools/pipeline_sweep.py 200 21 ) > $O/pipeline_sweep.txt 2>&1; echo "sweep rc=$?" >> $O/rc.txt; tail -2 $O/pipeline_sweep.txt
for v in 1 0 1; do
	( timeout 300 env PROBE_ONLY=lf_alone J40HIP_LF_RAW=$v python tools/r05_probe.py 256 16 3 ) >> $O/lf_alone_raw$v.jsonl 2>> $O/probe.err; echo "lf_alone raw=$v rc=$?" >> $O/rc.txt
done
( timeout 300 env PROBE_ONLY=device python tools/r05_probe.py 256 16 12 ) >> $O/device.jsonl 2>> $O/probe.err; echo "device rc=$?" >> $O/rc.txt
( cd /tmp && timeout 300 env PROBE_ONLY=lf_alone rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/tools/r05_probe.py 256 16 2 ) > $O/trace.txt 2>&1; echo "trace rc=$?" >> $O/rc.txt
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp "$f" $O/lf_alone_kernel_stats.csv; grep -E "k_lf_|Name" $O/lf_alone_kernel_stats.csv
find $O/trace -name '*.db' -delete; find $O/trace -name '*kernel_trace.csv' -delete
cat $O/rc.txt; for f in $O/lf_alone_raw0.jsonl $O/lf_alone_raw1.jsonl $O/device.jsonl; do echo $f; cut -c1-700 $f; done
