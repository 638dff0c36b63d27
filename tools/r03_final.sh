# the round's closing evidence: GPU tests, the default bench line, kernel stats of the timed region
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/gputest_final_r03.txt; cat gpurun_out/gputest_final_r03.txt
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $R/gpurun_out/bench_final_r03.json 2> $R/gpurun_out/bench_final_r03.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_final -- python $R/bench.py --skip-sections --no-cpu-baseline --steps 6 --warmup 2 > $R/gpurun_out/kt_final.log 2>&1
cd $R
python tools/prof_summary.py gpurun_out/kt_final gpurun_out/kernel_stats_final_r03.txt > /dev/null 2>&1; rm -rf gpurun_out/kt_final
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_final_r03.json"))
print({k: d[k] for k in ("value", "ms_per_step", "steps", "warmup")}); r = d["roofline"]; print({k: r[k] for k in r if k not in ("note", "traffic_source")}); print({k: v for k, v in d["pipeline"].items() if k != "note"})
PY
head -9 gpurun_out/kernel_stats_final_r03.txt | cut -c1-170
