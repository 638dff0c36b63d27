#!/bin/bash
# LfGroup launches in flight (hardware queues active at once) against the copies back and against the device's own pace
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04j; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
B="--skip-sections --no-cpu-baseline --warmup 2 --steps 20"
for fl in 2 1 4; do
  J40HIP_LF_FLIGHTS=$fl J40HIP_ASYNC_TIMING=1 timeout 900 python bench.py $B > $O/h2h_steps20_flights$fl.json 2> $O/h2h_steps20_flights$fl.err
done
J40HIP_LF_FLIGHTS=4 J40HIP_LF_FLIGHT_FRAMES=512 timeout 900 python bench.py $B > $O/h2h_steps20_flights4_512_as_before.json 2> $O/h2h_steps20_flights4_512_as_before.err
for fl in 2 1 4; do
  J40HIP_LF_FLIGHTS=$fl timeout 400 python tools/device_output_probe.py 256 12 device 2 > $O/dev_256_flights$fl.json 2> $O/dev_256_flights$fl.err
done
J40HIP_LF_FLIGHTS=4 J40HIP_LF_FLIGHT_FRAMES=512 timeout 400 python tools/device_output_probe.py 256 12 device 2 > $O/dev_256_flights4_512_as_before.json 2> $O/dev_256_flights4_512_as_before.err
ls $O
