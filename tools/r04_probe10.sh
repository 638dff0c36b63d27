#!/bin/bash
# pinned staging buffers per worker thread (6 against round 3's up to 64) on the long host-to-host runs, and on the device's own pace
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04k; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
B="--skip-sections --no-cpu-baseline --warmup 2"
J40HIP_ASYNC_TIMING=1 timeout 900 python bench.py $B --steps 20 > $O/h2h_steps20_cap6.json 2> $O/h2h_steps20_cap6.err
J40HIP_STAGE_BUFFERS=64 J40HIP_ASYNC_TIMING=1 timeout 900 python bench.py $B --steps 20 > $O/h2h_steps20_cap64.json 2> $O/h2h_steps20_cap64.err
timeout 900 python bench.py $B --steps 12 > $O/h2h_steps12_cap6.json 2> $O/h2h_steps12_cap6.err
J40HIP_STAGE_BUFFERS=3 timeout 900 python bench.py $B --steps 20 > $O/h2h_steps20_cap3.json 2> $O/h2h_steps20_cap3.err
timeout 400 python tools/device_output_probe.py 256 12 device 2 > $O/dev_256_cap6.json 2> $O/dev_256_cap6.err
J40HIP_STAGE_BUFFERS=64 timeout 400 python tools/device_output_probe.py 256 12 device 2 > $O/dev_256_cap64.json 2> $O/dev_256_cap64.err
ls $O
