#!/bin/bash
# round 6, call I: where one call through the public API spends its 29 ms (J40HIP_API_TIMING: parse phases, upload phases)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06i; mkdir -p $O
python - <<'PY' > $O/synth.log 2>&1
import sys; sys.path.insert(0, "tests")
from streams import synth
for i in range(4): synth("vardct", 7680, 4320, 3 + 1000 * i, forward=1)
PY
P8K=$(ls $R/build/streams/vardct_7680_4320_*forward-1.jxl | head -4 | tr '\n' ' ')
for run in 1 2; do
J40HIP_PLAN_TIMING=1 J40HIP_API_TIMING=1 J40HIP_SERVE=0 timeout 200 $R/build/api_threads 1 8 --warm 2 $P8K > $O/api_$run.json 2> $O/api_$run.err; echo "api rc=$?" >> $O/rc.txt
done
cat $O/api_1.json; tail -24 $O/api_1.err; nproc; lscpu | grep -E "Model name" ; cat /sys/fs/cgroup/cpu.max 2>/dev/null

