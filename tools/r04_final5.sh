#!/bin/bash
# the GPU suite on the final commit, and the API harness at the serving defaults
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04p; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/gputest.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
P8K=$(ls $R/build/streams/vardct_7680_4320_*forward-1.jxl | head -4 | tr '\n' ' ')
for rep in 1 2; do timeout 300 $R/build/api_threads 64 8 --warm 3 --verify-every 8 $P8K > $O/api_64_threads_$rep.json 2> $O/api_64_threads_$rep.err; done
timeout 300 $R/build/api_threads 64 8 --warm 3 $P8K > $O/api_64_threads_verify_all.json 2> $O/api_64_threads_verify_all.err
cat $O/gputest.txt; tail -2 $O/smoke.txt; for f in $O/api_*.json; do cut -c1-260 $f; done
