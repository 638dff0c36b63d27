#!/bin/bash
# Round 4's closing evidence, second half (after tools/r04_final.sh landed on a box whose host side was busy: `value` 8.3): the default
# bench line again, kernel traces of the pipeline in the three regimes bench.py reports k_hf_lanes in -- the timed region (pixels copied
# back: the copies are blit kernels that share the device with it), the pixels left in HBM, one batch alone -- and the API harness.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04h; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 1500 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
T="--skip-sections --no-cpu-baseline --steps 6 --warmup 2"
J40HIP_ASYNC_TIMING=1 timeout 600 python $R/bench.py $T > $O/bench_timed_region_only.json 2> $O/bench_timed_region_only.err
rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/bench.py $T > $O/kt.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt_devout -- python $R/tools/device_output_probe.py 256 6 device 2 > $O/kt_devout.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt_alone -- python $R/tools/device_output_probe.py 256 4 host 1 > $O/kt_alone.log 2>&1
P8K=$(ls $R/build/streams/vardct_7680_4320_*forward-1.jxl | head -4 | tr '\n' ' ')
for rep in 1 2; do timeout 300 $R/build/api_threads 64 8 --warm 3 --verify-every 8 $P8K > $O/api_64_threads_$rep.json 2> $O/api_64_threads_$rep.err; done
timeout 300 $R/build/api_threads 64 8 --warm 3 $P8K > $O/api_64_threads_verify_all.json 2> $O/api_64_threads_verify_all.err
timeout 300 $R/build/api_threads 128 8 --warm 3 --verify-every 8 $P8K > $O/api_128_threads.json 2> $O/api_128_threads.err
J40HIP_API_TIMING=1 J40HIP_SERVE=0 timeout 300 $R/build/api_threads 1 8 --warm 2 $P8K > $O/api_one_thread_latency.json 2> $O/api_one_thread_latency.err
cd $R
python tools/prof_summary.py $O/kt $O/kernel_stats_timed_region.txt > /dev/null 2>&1
python tools/prof_summary.py $O/kt_devout $O/kernel_stats_device_output_b256.txt > /dev/null 2>&1
python tools/prof_summary.py $O/kt_alone $O/kernel_stats_one_batch_alone_b256.txt > /dev/null 2>&1
rm -rf $O/kt $O/kt_devout $O/kt_alone
ls $O
