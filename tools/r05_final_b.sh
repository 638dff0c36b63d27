#!/bin/bash
# Round 5's closing evidence, second part, on the round's last commit (the kernels are those of tools/r05_final.sh's pass; bench.py
# has gained fields -- device_output.steady's neighbour pipeline.cgroup_cpu_in_region -- and the suite two tests since): the new GPU
# tests, then both bench lines again
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build_b.log 2>&1
cat $R/.commit_for_profiles > $O/commit_bench_lines.txt 2>/dev/null
timeout 600 python -u -m pytest tests/test_device_stages.py -m gpu -q -p no:cacheprovider -k "earlier_forms or damaged or 7680" > $O/gputest_new_tests.txt 2>&1; echo "gputest_new rc=$? $(tail -n 1 $O/gputest_new_tests.txt)" >> $O/rc_b.txt
timeout 600 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench_default rc=$?" >> $O/rc_b.txt
timeout 600 python $R/bench.py --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench_steps20_warmup5.err; echo "bench_steps20 rc=$?" >> $O/rc_b.txt
cat $O/rc_b.txt; tail -n 3 $O/gputest_new_tests.txt
python - <<'PY'
import json
for f in ("bench_default", "bench_steps20_warmup5"):
    r = json.loads(open("gpurun_out/r05/%s.json" % f).read().strip().splitlines()[-1])
    print(f, r["value"], r["ms_per_step"], r["pcie"]["achieved_gb_per_s"], r["pipeline"]["cgroup_cpu_in_region"], "device_output", r["device_output"]["ms_per_step"], r["device_output"]["steady"])
PY
