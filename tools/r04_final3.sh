#!/bin/bash
# Round 4's closing evidence, third part: the default bench line with the LfGroup streams decided per frame (`auto`: the lane decoder's
# launches beside the copies back were what made long host-to-host runs slow), at the default steps and at the driver's; the kernel trace of
# the timed region's command.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04l; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 1500 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
J40HIP_ASYNC_TIMING=1 timeout 900 python $R/bench.py --skip-sections --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_steps20_warmup5_timed_region.json 2> $O/bench_steps20_warmup5_timed_region.err
rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/bench.py --skip-sections --no-cpu-baseline --steps 6 --warmup 2 > $O/kt.log 2>&1
cd $R
python tools/prof_summary.py $O/kt $O/kernel_stats_timed_region.txt > /dev/null 2>&1
rm -rf $O/kt
P8K=$(ls $R/build/streams/vardct_7680_4320_*forward-1.jxl | head -4 | tr '\n' ' ')
for rep in 1 2; do timeout 300 $R/build/api_threads 64 8 --warm 3 --verify-every 8 $P8K > $O/api_64_threads_$rep.json 2> $O/api_64_threads_$rep.err; done
timeout 300 $R/build/api_threads 128 8 --warm 3 --verify-every 8 $P8K > $O/api_128_threads.json 2> $O/api_128_threads.err
ls $O
