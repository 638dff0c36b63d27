#!/bin/bash
# round 5, call F: the public API under load on a fresh box (64 / 128 callers; LfGroup streams of the served frames on the host
# threads or on the device), the file path (streamed read + parse) through the reference's own CLI and the thread harness
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05f; mkdir -p $O
python -c "from tests.streams import synth" 2>/dev/null
python - <<'PY' > $O/synth.log 2>&1
import sys; sys.path.insert(0, "tests")
from streams import synth
for i in range(4): synth("vardct", 7680, 4320, 3 + 1000 * i, forward=1)
PY
P8K=$(ls build/streams/vardct_7680_4320_*forward-1.jxl | head -4 | tr '\n' ' ')
run() { name=$1; shift; ( timeout 200 env "$@" ) > $O/$name.json 2> $O/$name.err; echo "$name rc=$? $(tail -n 1 $O/$name.json | cut -c1-330)" >> $O/rc.txt; }
run api64 A=1 build/api_threads 64 8 --warm 3 --verify-every 8 $P8K
run api64_lfdev J40HIP_SERVE_LF=device build/api_threads 64 8 --warm 3 --verify-every 8 $P8K
run api128 A=1 build/api_threads 128 8 --warm 3 --verify-every 8 $P8K
run api128_lfdev J40HIP_SERVE_LF=device build/api_threads 128 8 --warm 3 --verify-every 8 $P8K
run api64_b A=1 build/api_threads 64 8 --warm 3 --verify-every 8 $P8K
run api1 J40HIP_API_TIMING=1 J40HIP_SERVE=0 build/api_threads 1 8 --warm 2 $P8K
run api1_nostream J40HIP_API_TIMING=1 J40HIP_SERVE=0 J40HIP_STREAM=0 build/api_threads 1 8 --warm 2 $P8K
timeout 300 python -u -m pytest tests/test_api_threads.py tests/test_gpu_parity.py -q -x -m gpu -k "not 64_threads" > $O/tests.txt 2>&1; echo "tests rc=$? $(tail -n 1 $O/tests.txt)" >> $O/rc.txt
cat $O/rc.txt; tail -n 12 $O/api1.err | cut -c1-200; tail -n 4 $O/api1_nostream.err | cut -c1-200
