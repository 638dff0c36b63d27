#!/bin/bash
# round 4, second GPU call: whole GPU suite, serving sweep with a proper warm-up, host-to-host bench variants, RCCL dry run
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04b; mkdir -p $O
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
J40HIP_API_TIMING=1 J40HIP_SERVE=0 timeout 300 ./build/api_threads 1 6 --warm 2 $P8K > $O/api_latency.json 2> $O/api_latency.err
for cfg in "64 64 3 0" "64 64 6 0" "64 64 8 0" "64 64 6 1" "64 64 8 2" "64 32 8 1" "64 128 4 0" "128 64 8 0" "16 64 8 1"; do set -- $cfg
  J40HIP_SERVE_BATCH=$2 J40HIP_SERVE_IN_FLIGHT=$3 J40HIP_SERVE_WAIT_MS=$4 timeout 300 ./build/api_threads $1 8 --warm 3 $P8K > $O/api_t$1_b$2_f$3_w$4.json 2> $O/api_t$1_b$2_f$3_w$4.err
done
for cfg in "device 256 2" "host 256 2" "auto 256 2" "host 128 3" "host 64 4"; do set -- $cfg
  timeout 600 python bench.py --skip-sections --no-cpu-baseline --steps 12 --warmup 2 --lf-streams $1 --pipe-batch $2 --in-flight $3 > $O/bench_$1_$2_$3.json 2> $O/bench_$1_$2_$3.err
done
J40HIP_ASYNC_TIMING=1 timeout 600 python bench.py --skip-sections --no-cpu-baseline --steps 6 --warmup 1 --lf-streams host > $O/bench_timing.json 2> $O/bench_timing.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_dry_run.py > $O/rccl_dry_run.json 2> $O/rccl_dry_run.err
ls -la $O
