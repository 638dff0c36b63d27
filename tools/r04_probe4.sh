#!/bin/bash
# round 4, fourth GPU call: the fast latency kernel; why the host-to-host steps vary (copy-stream markers vs hardware queues); the device's pace at 256 / 512 frames per launch
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04d; mkdir -p $O
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
J40HIP_API_TIMING=1 J40HIP_SERVE=0 timeout 300 ./build/api_threads 1 6 --warm 2 $P8K > $O/api_latency_fast.json 2> $O/api_latency_fast.err
J40HIP_K1_FAST=0 J40HIP_API_TIMING=1 J40HIP_SERVE=0 timeout 300 ./build/api_threads 1 6 --warm 2 $P8K > $O/api_latency_general.json 2> $O/api_latency_general.err
B="--skip-sections --no-cpu-baseline --steps 8 --warmup 2 --pipe-batch 256 --in-flight 2"
run() { name=$1; shift; env "$@" timeout 600 python bench.py $B --lf-streams $LF > $O/h2h_${LF}_$name.json 2> $O/h2h_${LF}_$name.err; }
for LF in host device; do
  run default J40HIP_ASYNC_TIMING=1
  run groups1 J40HIP_COPY_GROUPS=1
  run side3 J40HIP_SIDE_STREAMS=3
  run copylow J40HIP_COPY_PRIORITY=low
done
LF=host run hwq8 GPU_MAX_HW_QUEUES=8
LF=auto run default J40HIP_ASYNC_TIMING=1
LF=auto run side3 J40HIP_SIDE_STREAMS=3
timeout 400 python tools/device_output_probe.py 256 4 device 2 > $O/dev_256_f2.json 2> $O/dev_256_f2.err
timeout 400 python tools/device_output_probe.py 512 3 device 2 > $O/dev_512_f2.json 2> $O/dev_512_f2.err
timeout 400 python tools/device_output_probe.py 512 3 device 1 > $O/dev_512_f1.json 2> $O/dev_512_f1.err
J40HIP_STREAM_LAYOUT=2 timeout 400 python tools/device_output_probe.py 512 3 device 2 > $O/dev_512_layout2.json 2> $O/dev_512_layout2.err
J40HIP_K1_QUEUE_WAVES=0 timeout 400 python tools/device_output_probe.py 512 3 device 2 > $O/dev_512_static.json 2> $O/dev_512_static.err
timeout 400 python tools/device_output_probe.py 256 4 host 2 > $O/dev_256_host.json 2> $O/dev_256_host.err
for cfg in "64 70" "64 100" "64 150" "128 100" "32 70"; do set -- $cfg
  J40HIP_SERVE_WAIT_MS=$2 timeout 300 ./build/api_threads $1 8 --warm 3 --verify-every 8 $P8K > $O/api_t$1_w$2.json 2> $O/api_t$1_w$2.err
done
ls $O | wc -l
