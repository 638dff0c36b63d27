#!/bin/bash
# round 6, call M: the single-image path in two phases (runtime.hip, decode_two_phase) -- its GPU test, the single-image path's parity suite,
# the public API's latency with and without (tools/latency_probe.py), api_threads (dj40.c's sequence, pixels checked against the reference)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06m; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_phases" ) > $O/pytest_two.txt 2>&1; echo "two-phase test rc=$?" >> $O/rc.txt; tail -15 $O/pytest_two.txt
for rep in 1 2; do for v in 1 0; do
	( timeout 300 env J40HIP_TWO_PHASE=$v python tools/latency_probe.py 9 ) >> $O/latency_two_phase_$v.jsonl 2>> $O/probe.err; echo "latency two_phase=$v rc=$?" >> $O/rc.txt
done; done
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_forward_streams.py tests/test_api_threads.py -m gpu -x -q ) > $O/pytest.txt 2>&1; echo "tests rc=$?" >> $O/rc.txt
tail -3 $O/pytest.txt
cat $O/rc.txt; for v in 1 0; do echo two_phase=$v; cut -c1-500 $O/latency_two_phase_$v.jsonl; done
