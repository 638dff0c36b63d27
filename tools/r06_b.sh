#!/bin/bash
# round 6, call B: (1) every SDMA engine by itself, both directions (tools/ubench/copy_probe way 11); (2) the driver's command first thing
# on the fresh box, then the timed pipeline alone at the driver's 20 steps, five times, each with the link probe on the line and the
# pipeline's own per-batch copy times; (3) one short run with the runtime's copy log (which engine a copy is given).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06b; mkdir -p $O
sys() { { echo "== $1"; cat /proc/buddyinfo; grep -E "MemFree|MemAvailable|AnonHuge|HugePages_Total|Hugepagesize|Unevictable|Mlocked" /proc/meminfo; cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag 2>/dev/null; ls /sys/class/iommu 2>/dev/null | head -3; } >> $O/sys.txt 2>&1; }
sys start
( timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/driver_like.json 2> $O/driver_like.err; echo "driver_like rc=$?" >> $O/rc.txt
sys after_driver_like
( timeout 120 build/copy_probe 16 4 0x800 ) > $O/engines.txt 2>&1; echo "engines rc=$?" >> $O/rc.txt
for i in 1 2 3 4 5; do
	( timeout 200 env J40HIP_ASYNC_TIMING=1 python bench.py --skip-sections --no-cpu-baseline --steps 20 --warmup 5 ) >> $O/steps20.jsonl 2> $O/steps20_$i.err; echo "steps20 $i rc=$?" >> $O/rc.txt
	grep "j40hip batch" $O/steps20_$i.err | awk '{print $(NF-1)}' | tr '\n' ' ' > $O/steps20_$i.copyms.txt; grep -v "j40hip batch" $O/steps20_$i.err | tail -20 > $O/steps20_$i.tail; rm -f $O/steps20_$i.err
	sys after_steps20_$i
done
( timeout 200 env AMD_LOG_LEVEL=4 AMD_LOG_MASK=0x300 python bench.py --skip-sections --no-cpu-baseline --steps 2 --warmup 1 --batch 64 --pipe-batch 64 ) > $O/copylog.json 2> $O/copylog.err; echo "copylog rc=$?" >> $O/rc.txt
grep -c "" $O/copylog.err >> $O/rc.txt; grep -i "HSA Copy\|copy_engine\|blit" $O/copylog.err | cut -c1-260 | sort | uniq -c | sort -rn | head -40 > $O/copylog_summary.txt; head -c 20000 $O/copylog.err > $O/copylog_head.txt; rm -f $O/copylog.err
cat $O/rc.txt; cat $O/engines.txt
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06b"
for f in ("driver_like.json", "steps20.jsonl"):
    for l in open(O + "/" + f):
        if not l.startswith("{"): continue
        r = json.loads(l); print(f, r["value"], r["ms_per_step"], r["pcie"]["achieved_gb_per_s"], r["pcie"].get("slow_run"), r["pcie"].get("link_probe"), r["pipeline"]["cgroup_cpu_in_region"], r["pipeline"]["host_stage_ms_per_frame"])
PY
for i in 1 2 3 4 5; do echo "run $i copy ms per batch:"; cat $O/steps20_$i.copyms.txt; echo; done
cat $O/copylog_summary.txt | head -20
tail -30 $O/sys.txt
