#!/bin/bash
# round 4, closing GPU call: k_vardct_large with 512 lanes -- parity in both forms, the pixel stage on the maxlog-8 stream, kernel stats;
# the public API under threads and the smoke of the final commit. Writes gpurun_out/r04L2/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04L2; mkdir -p $O
timeout 120 python -m pytest tests/test_pipeline.py -q -x -m gpu -k "large_transforms" -s > $O/test_large.txt 2>&1; echo "large test rc=$?" | tee -a $O/rc.txt
J40HIP_LARGE_IDCT=sweeps timeout 120 python -m pytest tests/test_pipeline.py -q -x -m gpu -k "large_transforms" -s > $O/test_large_sweeps.txt 2>&1; echo "large test (round 3's kernel) rc=$?" | tee -a $O/rc.txt
timeout 60 python tools/large_probe.py 64 3 16 > $O/probe_levels.json 2> $O/probe_levels.err; echo "probe rc=$?" | tee -a $O/rc.txt
J40HIP_LARGE_IDCT=sweeps timeout 60 python tools/large_probe.py 64 3 16 > $O/probe_sweeps.json 2> $O/probe_sweeps.err; echo "probe sweeps rc=$?" | tee -a $O/rc.txt
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/tools/large_probe.py 64 2 16 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1); echo "rocprof rc=$?" | tee -a $O/rc.txt
python tools/rocpd_stats.py $(ls $O/prof/*/*_results.db | head -1) > $O/kernel_stats.txt 2>&1
timeout 150 python -m pytest tests/test_gpu_parity.py tests/test_pipeline.py tests/test_api_threads.py -q -x -m gpu -k "all_transforms or maxlog or batches_give or mix_of_images or group_ranges" > $O/test_more.txt 2>&1; echo "more tests rc=$?" | tee -a $O/rc.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt
tail -n 3 $O/test_large.txt; cat $O/probe_levels.json $O/probe_sweeps.json; head -5 $O/kernel_stats.txt; tail -n 2 $O/smoke.txt; tail -n 3 $O/test_more.txt
