#!/bin/bash
# round 6, call H: k_lf_rows with leaf-only channels left as residuals (k_lf_predict) -- the device-stage and pipeline tests, then the launch
# alone on the device with and without (PROBE_ONLY=lf_alone), then the pixels-in-HBM pipeline both ways
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06h; mkdir -p $O
( timeout 1500 python -m pytest tests/test_device_stages.py tests/test_pipeline.py tests/test_lf_decoder_glue.py -m gpu -x -q ) > $O/pytest.txt 2>&1; echo "tests rc=$?" >> $O/rc.txt
tail -5 $O/pytest.txt
for v in 0 1 0 1; do
	( timeout 300 env PROBE_ONLY=lf_alone J40HIP_LF_RAW=$v python tools/r05_probe.py 256 16 3 ) >> $O/lf_alone_raw$v.jsonl 2>> $O/probe.err; echo "lf_alone raw=$v rc=$?" >> $O/rc.txt
done
for v in 0 1; do
	( timeout 300 env PROBE_ONLY=device J40HIP_LF_RAW=$v python tools/r05_probe.py 256 16 12 ) >> $O/device_raw$v.jsonl 2>> $O/probe.err; echo "device raw=$v rc=$?" >> $O/rc.txt
done
( cd /tmp && timeout 300 env PROBE_ONLY=lf_alone rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/tools/r05_probe.py 256 16 2 ) > $O/trace.txt 2>&1; echo "trace rc=$?" >> $O/rc.txt
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp "$f" $O/lf_alone_kernel_stats.csv; grep -E "k_lf_|Name" $O/lf_alone_kernel_stats.csv
find $O/trace -name '*.db' -delete; find $O/trace -name '*kernel_trace.csv' -delete
cat $O/rc.txt; for f in $O/lf_alone_raw0.jsonl $O/lf_alone_raw1.jsonl $O/device_raw0.jsonl $O/device_raw1.jsonl; do echo $f; cut -c1-600 $f; done
