#!/bin/bash
# round 6, call C: the copies back on the measured SDMA engine (hostcopy.hip) against hipMemcpyAsync (J40HIP_COPY_ENGINE=hip), the timed
# pipeline alone at the driver's 20 steps, alternating; the pipeline's GPU tests first
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06c; mkdir -p $O
( timeout 900 python -m pytest tests/test_pipeline.py tests/test_api_threads.py -m gpu -x -q ) > $O/pytest_pipeline.txt 2>&1; echo "pytest rc=$?" >> $O/rc.txt
for i in 1 2 3 4 5; do
	( timeout 200 env J40HIP_ASYNC_TIMING=1 python bench.py --skip-sections --no-cpu-baseline --steps 20 --warmup 5 ) >> $O/sdma.jsonl 2> $O/sdma_$i.err; echo "sdma $i rc=$?" >> $O/rc.txt
	grep "j40hip batch" $O/sdma_$i.err | awk '{print $(NF-1)}' | tr '\n' ' ' > $O/sdma_$i.copyms.txt; grep "hostcopy" $O/sdma_$i.err > $O/sdma_$i.engine.txt; grep -v "j40hip batch" $O/sdma_$i.err | tail -12 > $O/sdma_$i.tail; rm -f $O/sdma_$i.err
	if [ $i -le 2 ]; then
		( timeout 200 env J40HIP_COPY_ENGINE=hip J40HIP_ASYNC_TIMING=1 python bench.py --skip-sections --no-cpu-baseline --steps 20 --warmup 5 ) >> $O/hip.jsonl 2> $O/hip_$i.err; echo "hip $i rc=$?" >> $O/rc.txt
		grep "j40hip batch" $O/hip_$i.err | awk '{print $(NF-1)}' | tr '\n' ' ' > $O/hip_$i.copyms.txt; rm -f $O/hip_$i.err
	fi
done
cat $O/rc.txt; tail -3 $O/pytest_pipeline.txt
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06c"
for f in ("sdma.jsonl", "hip.jsonl"):
    for l in open(O + "/" + f):
        if not l.startswith("{"): continue
        r = json.loads(l); print(f, r["value"], r["ms_per_step"], r["pcie"]["achieved_gb_per_s"], r["pcie"].get("slow_run"), r["pcie"].get("copy_engine"), r["pipeline"]["cgroup_cpu_in_region"], r["pipeline"]["host_stage_ms_per_frame"])
PY
for i in 1 2 3 4 5; do echo "sdma run $i copy ms per batch:"; cat $O/sdma_$i.copyms.txt; echo; cat $O/sdma_$i.engine.txt; done
for i in 1 2; do echo "hip run $i copy ms per batch:"; cat $O/hip_$i.copyms.txt; echo; done
