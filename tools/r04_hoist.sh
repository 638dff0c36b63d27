#!/bin/bash
# round 4: the pixel kernels' frame state hoisted out of the tile loop -- parity through the batch-wide launches, then the stages alone
# (and the variant that also zeroes the tiles under the record loads). Writes gpurun_out/r04H/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04H; mkdir -p $O
timeout 100 python -m pytest tests/test_pipeline.py -q -x -m gpu -k "batches_give or large_transforms or forward_encoded_8k or matches_single_image" > $O/tests.txt 2>&1; echo "tests rc=$?" | tee -a $O/rc.txt
timeout 40 python tools/stages_alone_probe.py 256 3 8 >> $O/alone_hoisted.json 2>> $O/err.txt; echo "probe rc=$?" | tee -a $O/rc.txt
J40HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libj40hip_overlap.so timeout 40 python tools/stages_alone_probe.py 256 3 8 >> $O/alone_overlap.json 2>> $O/err.txt; echo "probe overlap rc=$?" | tee -a $O/rc.txt
timeout 40 python tools/stages_alone_probe.py 256 3 8 >> $O/alone_hoisted.json 2>> $O/err.txt; echo "probe rc=$?" | tee -a $O/rc.txt
tail -n 3 $O/tests.txt; cat $O/alone_hoisted.json $O/alone_overlap.json
