#!/bin/bash
# round 5, call G: the serving pipeline's "auto" choice of who decodes the LfGroup streams (64 / 128 / 32 callers), then the API tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05g; mkdir -p $O
python - <<'PY' > $O/synth.log 2>&1
import sys; sys.path.insert(0, "tests")
from streams import synth
for i in range(4): synth("vardct", 7680, 4320, 3 + 1000 * i, forward=1)
PY
P8K=$(ls build/streams/vardct_7680_4320_*forward-1.jxl | head -4 | tr '\n' ' ')
run() { name=$1; shift; ( timeout 200 env "$@" ) > $O/$name.json 2> $O/$name.err; echo "$name rc=$? $(tail -n 1 $O/$name.json | cut -c1-330)" >> $O/rc.txt; }
run api64 A=1 build/api_threads 64 8 --warm 3 --verify-every 8 $P8K
run api128 A=1 build/api_threads 128 8 --warm 3 --verify-every 8 $P8K
run api32 A=1 build/api_threads 32 8 --warm 3 --verify-every 8 $P8K
run api128_b A=1 build/api_threads 128 8 --warm 3 --verify-every 8 $P8K
run api96 A=1 build/api_threads 96 8 --warm 3 --verify-every 8 $P8K
timeout 400 python -u -m pytest tests/test_api_threads.py -q -x -m gpu > $O/tests.txt 2>&1; echo "tests rc=$? $(tail -n 1 $O/tests.txt)" >> $O/rc.txt
cat $O/rc.txt
