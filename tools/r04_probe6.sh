#!/bin/bash
# round 4, sixth GPU call: the branch-light fast latency kernel under the whole suite, single-call latency, the full default bench line
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04f; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
python - > $O/synth.log 2>&1 <<'PY'
import sys, concurrent.futures
sys.path.insert(0, "tests")
from streams import synth
with concurrent.futures.ThreadPoolExecutor(16) as ex:
    list(ex.map(lambda i: synth("vardct", 7680, 4320, 3 + 1000 * i, forward=1), range(64)))
PY
timeout 1500 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt
P8K=$(ls build/streams/vardct_7680_4320_*forward-1.jxl | head -4 | tr '\n' ' ')
J40HIP_API_TIMING=1 J40HIP_SERVE=0 timeout 300 ./build/api_threads 1 8 --warm 2 $P8K > $O/api_latency.json 2> $O/api_latency.err
timeout 300 ./build/api_threads 64 8 --warm 3 --verify-every 8 $P8K > $O/api_t64.json 2> $O/api_t64.err
timeout 300 ./build/api_threads 64 8 --warm 3 $P8K > $O/api_t64_verify_all.json 2> $O/api_t64_verify_all.err
timeout 1500 python bench.py > $O/bench_full.json 2> $O/bench_full.err
ls $O | wc -l
