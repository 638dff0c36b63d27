#!/bin/bash
# round 6, call F: the LDS-tiled EPF kernels -- parity tests, then their durations
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06f; mkdir -p $O
( timeout 1200 python -m pytest tests/test_restoration.py -m gpu -x -q ) > $O/pytest_restoration.txt 2>&1; echo "restoration rc=$?" >> $O/rc.txt
tail -5 $O/pytest_restoration.txt
( timeout 600 python tools/restore_probe.py 3 ) > $O/restore_probe.txt 2>&1; echo "probe rc=$?" >> $O/rc.txt
cat $O/restore_probe.txt
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/tools/restore_probe.py 3 ) > $O/trace.txt 2>&1; echo "trace rc=$?" >> $O/rc.txt
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp "$f" $O/restore_kernel_stats.csv; head -12 $O/restore_kernel_stats.csv
find $O/trace -name '*.db' -delete; find $O/trace -name '*kernel_trace.csv' -delete
cat $O/rc.txt
