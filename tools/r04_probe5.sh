#!/bin/bash
# round 4, fifth GPU call: the device's pace in steady state (10+ steps) at 256 / 384 / 512 frames per launch; host-LF with CPUs left for the runtime; config 5
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04e; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
python - > $O/synth.log 2>&1 <<'PY'
import sys, concurrent.futures
sys.path.insert(0, "tests")
from streams import synth
with concurrent.futures.ThreadPoolExecutor(16) as ex:
    list(ex.map(lambda i: synth("vardct", 7680, 4320, 3 + 1000 * i, forward=1), range(64)))
PY
for cfg in "256 12 device 2" "512 6 device 1" "512 6 device 2" "384 8 device 2" "256 12 device 3"; do set -- $cfg
  J40HIP_ASYNC_TIMING=1 timeout 400 python tools/device_output_probe.py $1 $2 $3 $4 > $O/dev_$1_f$4.json 2> $O/dev_$1_f$4.err
done
J40HIP_K1_QUEUE_WAVES=0 timeout 400 python tools/device_output_probe.py 512 6 device 2 > $O/dev_512_static.json 2> $O/dev_512_static.err
for cfg in "256 4 host 16" "256 4 host 12" "256 4 device 16" "256 4 auto 16" "512 2 host 12" "128 6 host 12"; do set -- $cfg
  J40HIP_ASYNC_TIMING=1 timeout 300 python tools/config5_probe.py $1 $2 $3 $4 > $O/c5_$1_f$2_$3_t$4.json 2> $O/c5_$1_f$2_$3_t$4.err
done
B="--skip-sections --no-cpu-baseline --steps 8 --warmup 2 --pipe-batch 256 --in-flight 2"
timeout 600 python bench.py $B --lf-streams host --host-threads 12 > $O/h2h_host_t12.json 2> $O/h2h_host_t12.err
timeout 600 python bench.py $B --lf-streams host --host-threads 8 > $O/h2h_host_t8.json 2> $O/h2h_host_t8.err
timeout 600 python bench.py $B --lf-streams auto --host-threads 12 > $O/h2h_auto_t12.json 2> $O/h2h_auto_t12.err
P8K=$(ls build/streams/vardct_7680_4320_*forward-1.jxl | head -4 | tr '\n' ' ')
for cfg in "64 100 16" "64 100 12" "64 100 10" "128 100 12" "64 0 12"; do set -- $cfg
  J40HIP_SERVE_WAIT_MS=$2 J40HIP_SERVE_THREADS=$3 timeout 300 ./build/api_threads $1 8 --warm 3 --verify-every 8 $P8K > $O/api_t$1_w$2_s$3.json 2> $O/api_t$1_w$2_s$3.err
done
ls $O | wc -l
