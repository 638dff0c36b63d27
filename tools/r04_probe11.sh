#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04m; mkdir -p $O
R=$PWD
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/kt -- python $R/bench.py --skip-sections --no-cpu-baseline --steps 14 --warmup 2 --lf-streams device > $R/$O/bench.json 2> $R/$O/bench.err
cd $R
python tools/copy_timeline.py $O/kt $O/copy_timeline.txt > /dev/null 2>&1
rm -rf $O/kt
cat $O/copy_timeline.txt | head -60
