#!/bin/bash
# round 6, call E: the restoration filters' GPU tests, then the whole GPU suite (the pipeline's copies and the pixel kernels' template changed)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06e; mkdir -p $O
( timeout 1200 python -m pytest tests/test_restoration.py -m gpu -x -q ) > $O/pytest_restoration.txt 2>&1; echo "restoration rc=$?" >> $O/rc.txt
tail -30 $O/pytest_restoration.txt
( timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_restoration.py ) > $O/pytest_gpu.txt 2>&1; echo "suite rc=$?" >> $O/rc.txt
tail -15 $O/pytest_gpu.txt
cat $O/rc.txt
