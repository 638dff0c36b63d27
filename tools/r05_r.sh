#!/bin/bash
# round 5, call R: the copy stream at high priority (its blit kernels ahead of the pixel kernels for wavefront slots) against the
# default, the timed pipeline alone, alternating -- the clock's slow runs sit at 0.8 x the fast ones' copy rate, which is what copies
# that stand still while a batch's kernels run would give
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05r; mkdir -p $O
for i in 1 2 3; do
	( timeout 100 env J40HIP_COPY_PRIORITY=high python bench.py --skip-sections --steps 8 --warmup 2 ) >> $O/high.jsonl 2>> $O/high.err; echo "high$i rc=$?" >> $O/rc.txt
	( timeout 100 python bench.py --skip-sections --steps 8 --warmup 2 ) >> $O/default.jsonl 2>> $O/default.err; echo "default$i rc=$?" >> $O/rc.txt
done
cat $O/rc.txt
python - <<'PY'
import json
for f in ("high", "default"):
    for l in open("gpurun_out/r05r/%s.jsonl" % f):
        if not l.startswith("{"): continue
        r = json.loads(l); print(f, r["value"], r["ms_per_step"], r["pcie"]["achieved_gb_per_s"], r["pipeline"]["cgroup_cpu_in_region"], r["pipeline"]["host_stage_ms_per_frame"])
PY
