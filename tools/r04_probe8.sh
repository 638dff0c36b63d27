#!/bin/bash
# why the host-to-host step grows with the number of steps queued at once (6-8 steps: 0.70 s, 12: 1.02 s, 20: 1.62 s on the same box)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04i; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
B="--skip-sections --no-cpu-baseline --warmup 2"
( while true; do rocm-smi --showmeminfo vram 2>/dev/null | grep "Used" | awk '{print systime(), $NF}'; sleep 1; done ) > $O/vram_20.txt &
MON=$!
J40HIP_ASYNC_TIMING=1 timeout 900 python bench.py $B --steps 20 > $O/h2h_steps20.json 2> $O/h2h_steps20.err
kill $MON
J40HIP_ASYNC_TIMING=1 J40HIP_LF_CAP=512 timeout 900 python bench.py $B --steps 20 > $O/h2h_steps20_lfcap512.json 2> $O/h2h_steps20_lfcap512.err
J40HIP_ASYNC_TIMING=1 timeout 900 python bench.py $B --steps 20 --lf-streams auto > $O/h2h_steps20_auto.json 2> $O/h2h_steps20_auto.err
J40HIP_ASYNC_TIMING=1 timeout 900 python bench.py $B --steps 12 > $O/h2h_steps12.json 2> $O/h2h_steps12.err
ls $O
