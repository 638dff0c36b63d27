#!/bin/bash
# First GPU call of the next round: what round 4 left unmeasured, in the order of what a failure would cost.
#   1. the whole GPU suite on the round's last commit (the threaded plan build and the hoisted frame state ran subsets only)
#   2. the single-call latency through the public API with its breakdown (J40HIP_API_TIMING): the plan build by the parse's team
#   3. the pixel stage's phase table re-taken after the hoisting (instrumented variant: tools/build_variant.sh k2phases kernels.hip -DJ40_K2_PHASES
#      BEFORE the call -- variants are built here, the GPU box only runs them)
#   4. bench.py at its defaults
# Writes gpurun_out/r05a/. About 7 GPU-minutes.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05a; mkdir -p $O
timeout 600 python -m pytest tests -q -x -m gpu > $O/gputest.txt 2>&1; echo "gpu suite rc=$?" | tee -a $O/rc.txt
python - > $O/streams.txt 2>&1 <<'PY'
import sys; sys.path.insert(0, "tests")
from streams import synth, CACHE
import os
synth("vardct", 7680, 4320, 3, forward=1)
print(os.path.join(CACHE, "vardct_7680_4320_3_forward-1.jxl"))
PY
S=$(tail -n 1 $O/streams.txt)
J40HIP_SERVE=0 J40HIP_API_TIMING=1 timeout 120 build/api_threads 1 12 --warm 3 $S > $O/api_one_thread.json 2> $O/api_one_thread_timing.txt; echo "api rc=$?" | tee -a $O/rc.txt
if [ -f build/variants/libj40hip_k2phases.so ]; then
	PROBE_K2_PHASES=1 J40HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libj40hip_k2phases.so timeout 90 python tools/stages_alone_probe.py 256 3 8 > $O/phases.jsonl 2> $O/phases.err; echo "phases rc=$?" | tee -a $O/rc.txt
fi
timeout 60 python tools/stages_alone_probe.py 256 4 8 > $O/alone.json 2>> $O/phases.err; echo "alone rc=$?" | tee -a $O/rc.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
tail -n 3 $O/gputest.txt; cat $O/api_one_thread.json; tail -n 4 $O/api_one_thread_timing.txt; cat $O/alone.json; head -c 600 $O/bench.json
