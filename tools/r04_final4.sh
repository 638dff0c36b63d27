#!/bin/bash
# Round 4's closing evidence, last part (the commit with four worker threads for the pipelines whose LfGroup streams the GPU decodes):
# the default bench line, the driver's steps, the kernel trace of the timed region's command, the API harness with 6 / 8 / 12 serving threads
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04o; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 1500 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
J40HIP_ASYNC_TIMING=1 timeout 900 python $R/bench.py --skip-sections --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_steps20_warmup5_timed_region.json 2> $O/bench_steps20_warmup5_timed_region.err
cat /sys/fs/cgroup/cpu.stat > $O/cpu_stat_after_benches.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/bench.py --skip-sections --no-cpu-baseline --steps 6 --warmup 2 > $O/kt.log 2>&1
cd $R
python tools/prof_summary.py $O/kt $O/kernel_stats_timed_region.txt > /dev/null 2>&1
rm -rf $O/kt
P8K=$(ls $R/build/streams/vardct_7680_4320_*forward-1.jxl | head -4 | tr '\n' ' ')
for t in 12 8 6 4; do J40HIP_SERVE_THREADS=$t timeout 300 $R/build/api_threads 64 8 --warm 3 --verify-every 8 $P8K > $O/api_64_threads_serve$t.json 2> $O/api_64_threads_serve$t.err; done
J40HIP_SERVE_THREADS=8 timeout 300 $R/build/api_threads 128 8 --warm 3 --verify-every 8 $P8K > $O/api_128_threads_serve8.json 2> $O/api_128_threads_serve8.err
J40HIP_SERVE_THREADS=8 J40HIP_SERVE_LF=device timeout 300 $R/build/api_threads 64 8 --warm 3 --verify-every 8 $P8K > $O/api_64_threads_serve8_lfdevice.json 2> $O/api_64_threads_serve8_lfdevice.err
ls $O
