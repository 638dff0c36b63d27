#!/bin/bash
# round 6, call P: the closing sweeps on the final code -- random option mixes through the product library against the reference (GPU),
# damaged streams through the pipeline with the LfGroup streams on the device against the single-image path, two phases against one
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06p; mkdir -p $O
FUZZ_FLIPS=0.3 timeout 2400 python tools/fuzz_parity.py 600 909 gpu > $O/fuzz_gpu_600.txt 2>&1; echo "fuzz rc=$?" >> $O/rc.txt; tail -2 $O/fuzz_gpu_600.txt
timeout 1500 python tools/pipeline_sweep.py 500 33 > $O/pipeline_sweep_500.txt 2>&1; echo "pipeline sweep rc=$?" >> $O/rc.txt; tail -1 $O/pipeline_sweep_500.txt
timeout 1200 python tools/two_phase_sweep.py 200 17 > $O/two_phase_sweep_200.txt 2>&1; echo "two-phase sweep rc=$?" >> $O/rc.txt; tail -1 $O/two_phase_sweep_200.txt
cat $O/rc.txt
