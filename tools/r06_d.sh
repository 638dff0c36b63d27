#!/bin/bash
# round 6, call D: copies back on the measured SDMA engine AND the workers' uploads on engines of their own, the timed pipeline alone at the
# driver's 20 steps, six runs; two with hipMemcpyAsync for both (J40HIP_COPY_ENGINE=hip); the driver's full command at the end
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06d; mkdir -p $O
for i in 1 2 3 4 5 6; do
	( timeout 200 env J40HIP_ASYNC_TIMING=1 python bench.py --skip-sections --no-cpu-baseline --steps 20 --warmup 5 ) >> $O/sdma.jsonl 2> $O/sdma_$i.err; echo "sdma $i rc=$?" >> $O/rc.txt
	grep "j40hip batch" $O/sdma_$i.err | awk '{print $(NF-1)}' | tr '\n' ' ' > $O/sdma_$i.copyms.txt; grep "hostcopy" $O/sdma_$i.err > $O/sdma_$i.engine.txt; grep "host stage\|gpu thread" $O/sdma_$i.err | tail -6 > $O/sdma_$i.tail; rm -f $O/sdma_$i.err
	if [ $i -le 2 ]; then
		( timeout 200 env J40HIP_COPY_ENGINE=hip J40HIP_ASYNC_TIMING=1 python bench.py --skip-sections --no-cpu-baseline --steps 20 --warmup 5 ) >> $O/hip.jsonl 2> $O/hip_$i.err; echo "hip $i rc=$?" >> $O/rc.txt
		grep "host stage\|gpu thread" $O/hip_$i.err | tail -6 > $O/hip_$i.tail; rm -f $O/hip_$i.err
	fi
done
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/driver_like.json 2> $O/driver_like.err; echo "driver_like rc=$?" >> $O/rc.txt
cat $O/rc.txt
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06d"
for f in ("sdma.jsonl", "hip.jsonl", "driver_like.json"):
    for l in open(O + "/" + f):
        if not l.startswith("{"): continue
        r = json.loads(l); print(f, r["value"], r["ms_per_step"], r["pcie"]["achieved_gb_per_s"], r["pcie"].get("slow_run"), r["pcie"].get("copy_engine", {}).get("engine"), r["pcie"].get("copy_engine", {}).get("upload_engines_mask"), r["pipeline"]["cgroup_cpu_in_region"], r["pipeline"]["host_stage_ms_per_frame"], r.get("device_output", {}).get("value"))
PY
for i in 1 2 3 4 5 6; do echo "sdma run $i:"; cat $O/sdma_$i.engine.txt; cat $O/sdma_$i.tail | cut -c1-330; done
for i in 1 2; do echo "hip run $i:"; cat $O/hip_$i.tail | cut -c1-330; done
