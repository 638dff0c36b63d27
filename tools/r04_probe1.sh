#!/bin/bash
# round 4, first GPU call: correctness of the reworked pipeline / serving API, then where the host-to-host clock stands
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04a; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
python - > $O/synth.log 2>&1 <<'PY'
import sys, concurrent.futures
sys.path.insert(0, "tests")
from streams import synth
with concurrent.futures.ThreadPoolExecutor(16) as ex:
    list(ex.map(lambda i: synth("vardct", 7680, 4320, 3 + 1000 * i, forward=1), range(64)))
PY
timeout 1200 python -m pytest tests/test_api_threads.py tests/test_pipeline.py -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt
timeout 120 python tools/pcie_probe.py > $O/pcie.txt 2>&1
P8K=$(ls build/streams/vardct_7680_4320_*forward-1.jxl | head -4 | tr '\n' ' ')
for cfg in "1 5 0" "1 5 1"; do set -- $cfg
  J40HIP_API_TIMING=1 J40HIP_SERVE=$3 timeout 300 ./build/api_threads $1 $2 --warm 1 $P8K > $O/api_t$1_serve$3.json 2> $O/api_t$1_serve$3.err
done
for cfg in "64 64 3 3" "64 32 3 3" "64 64 4 3" "64 128 2 3" "64 64 3 0" "16 64 3 3" "256 64 3 3"; do set -- $cfg
  J40HIP_SERVE_BATCH=$2 J40HIP_SERVE_IN_FLIGHT=$3 J40HIP_SERVE_WAIT_MS=$4 timeout 300 ./build/api_threads $1 4 --warm 1 $P8K > $O/api_t$1_b$2_f$3_w$4.json 2> $O/api_t$1_b$2_f$3_w$4.err
done
J40HIP_SERVE_LF=device timeout 300 ./build/api_threads 64 4 --warm 1 $P8K > $O/api_t64_lfdevice.json 2>&1
for cfg in "device 256 2" "host 256 2" "auto 256 2" "host 128 3" "host 64 4" "device 128 3"; do set -- $cfg
  timeout 600 python bench.py --skip-sections --no-cpu-baseline --steps 12 --warmup 2 --lf-streams $1 --pipe-batch $2 --in-flight $3 > $O/bench_$1_$2_$3.json 2> $O/bench_$1_$2_$3.err
done
J40HIP_ASYNC_TIMING=1 timeout 600 python bench.py --skip-sections --no-cpu-baseline --steps 6 --warmup 1 --lf-streams host > $O/bench_timing.json 2> $O/bench_timing.err
ls -la $O
