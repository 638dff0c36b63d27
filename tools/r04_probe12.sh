#!/bin/bash
# do the copies back depend on the host's CPU time? long host-to-host runs with 2 / 4 / 16 worker threads (LfGroup streams on the GPU)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04n; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
B="--skip-sections --no-cpu-baseline --warmup 2 --steps 20 --lf-streams device"
for t in 4 2 16; do
  J40HIP_ASYNC_TIMING=1 timeout 900 python bench.py $B --host-threads $t > $O/h2h_steps20_t$t.json 2> $O/h2h_steps20_t$t.err
done
cat /sys/fs/cgroup/cpu.stat > $O/cpu_stat_after.txt 2>&1
for f in $O/h2h_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d['pcie']['achieved_gb_per_s'])"; done
cat $O/cpu_stat_after.txt
