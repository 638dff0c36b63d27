#!/bin/bash
# round 5, call Q: the contract clock's slow runs (one in six: 36-39 GB/s of copies back instead of 52) beside the container's CPU
# accounting over the timed region (bench.py: pipeline.cgroup_cpu_in_region) -- four pipeline threads (the default) and two, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05q; mkdir -p $O
for i in 1 2 3 4; do
	( timeout 200 python bench.py --skip-sections --steps 10 --warmup 2 ) >> $O/threads4.jsonl 2>> $O/threads4.err; echo "t4_$i rc=$?" >> $O/rc.txt
	if [ $i -le 3 ]; then ( timeout 200 python bench.py --skip-sections --steps 10 --warmup 2 --host-threads 2 ) >> $O/threads2.jsonl 2>> $O/threads2.err; echo "t2_$i rc=$?" >> $O/rc.txt; fi
done
cat $O/rc.txt
python - <<'PY'
import json
for f in ("threads4", "threads2"):
    for l in open("gpurun_out/r05q/%s.jsonl" % f):
        if not l.startswith("{"): continue
        r = json.loads(l); print(f, r["value"], r["ms_per_step"], r["pcie"]["achieved_gb_per_s"], r["pipeline"]["cgroup_cpu_in_region"], r["pipeline"]["host_stage_ms_per_frame"])
PY
