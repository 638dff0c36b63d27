#!/bin/bash
# a second sample of the default bench line (another box) on the final commit
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04q; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py --skip-modular > $O/bench_default_second_sample.json 2> $O/bench_default_second_sample.err
timeout 600 python $R/bench.py --skip-sections --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_steps20_second_sample.json 2> $O/bench_steps20_second_sample.err
cat /sys/fs/cgroup/cpu.stat > $O/cpu_stat.txt 2>&1
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT", os.getcwd())+"/gpurun_out/r04q/"
for f in ("bench_default_second_sample.json","bench_steps20_second_sample.json"):
    try:
        d=json.load(open(O+f)); print(f, d["value"], d["ms_per_step"], d["pcie"]["achieved_gb_per_s"], d["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
