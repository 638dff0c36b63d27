#!/bin/bash
# round 4, seventh GPU call: the copies back are blit KERNELS (__amd_rocclr_copyBuffer in the kernel trace), so the hardware queue the copy
# stream lands in decides whether they flow. Variants, twice each: copy stream at high priority (a queue of its own beside the batch
# streams), three pixel-kernel streams (the fourth shares the copy stream's queue), both; SDMA asked for explicitly.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04g; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
python - > $O/synth.log 2>&1 <<'PY'
import sys, concurrent.futures
sys.path.insert(0, "tests")
from streams import synth
with concurrent.futures.ThreadPoolExecutor(16) as ex:
    list(ex.map(lambda i: synth("vardct", 7680, 4320, 3 + 1000 * i, forward=1), range(64)))
PY
B="--skip-sections --no-cpu-baseline --steps 8 --warmup 2"
run() { name=$1; shift; env "$@" timeout 600 python bench.py $B > $O/h2h_$name.json 2> $O/h2h_$name.err; }
for rep in 1 2; do
  run default_$rep J40HIP_ASYNC_TIMING=1
  run copyhigh_$rep J40HIP_COPY_PRIORITY=high J40HIP_ASYNC_TIMING=1
  run side3_$rep J40HIP_SIDE_STREAMS=3
  run copyhigh_side3_$rep J40HIP_COPY_PRIORITY=high J40HIP_SIDE_STREAMS=3
done
run sdma1 HSA_ENABLE_SDMA=1
run sdma0 HSA_ENABLE_SDMA=0
cd /tmp
HSA_ENABLE_SDMA=1 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/kt_sdma1 -- python $OLDPWD/bench.py --skip-sections --no-cpu-baseline --steps 3 --warmup 1 > $OLDPWD/$O/kt_sdma1.log 2>&1
cd $OLDPWD
python tools/prof_summary.py $O/kt_sdma1 $O/kernel_stats_sdma1.txt > /dev/null 2>&1; rm -rf $O/kt_sdma1
P8K=$(ls build/streams/vardct_7680_4320_*forward-1.jxl | head -4 | tr '\n' ' ')
for rep in 1 2; do
  timeout 300 ./build/api_threads 64 8 --warm 3 --verify-every 8 $P8K > $O/api_64_default_$rep.json 2> $O/api_64_default_$rep.err
  J40HIP_COPY_PRIORITY=high timeout 300 ./build/api_threads 64 8 --warm 3 --verify-every 8 $P8K > $O/api_64_copyhigh_$rep.json 2> $O/api_64_copyhigh_$rep.err
done
J40HIP_COPY_PRIORITY=high timeout 300 ./build/api_threads 128 8 --warm 3 --verify-every 8 $P8K > $O/api_128_copyhigh.json 2> $O/api_128_copyhigh.err
timeout 300 ./build/api_threads 128 8 --warm 3 --verify-every 8 $P8K > $O/api_128_default.json 2> $O/api_128_default.err
ls $O | wc -l
