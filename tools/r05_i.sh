#!/bin/bash
# round 5, call I: k_lf_rows with the plain step instantiated per combination of needs -- parity, then the launch alone and the pipeline
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05i; mkdir -p $O
free -g > $O/free.txt 2>&1; nproc >> $O/free.txt
timeout 300 python -u -m pytest tests/test_device_stages.py tests/test_pipeline.py -q -x -m gpu -k "not config5 and not large_transforms and not queued" > $O/tests.txt 2>&1; echo "tests rc=$? $(tail -n 1 $O/tests.txt)" >> $O/rc.txt
probe() { name=$1; shift; ( timeout 150 env "$@" python tools/r05_probe.py 256 16 6 ) >> $O/probes.jsonl 2>> $O/$name.err; echo "$name rc=$?" >> $O/rc.txt; }
probe lf_alone PROBE_ONLY=lf_alone
probe device PROBE_ONLY=device
probe lf_alone_b PROBE_ONLY=lf_alone
probe alone PROBE_ONLY=alone
cat $O/rc.txt; cat $O/free.txt; cat $O/probes.jsonl
