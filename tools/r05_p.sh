#!/bin/bash
# round 5, call P: does the NUMA node of the pinned landing buffers decide the copy-back rate? The contract clock is PCIe time; two
# evidence passes gave 13.2 Gpx/s (53 GB/s) and 9.0 (36 GB/s) for the same command minutes apart. The timed pipeline alone
# (--skip-sections), the process bound to the GPU's node (J40_BENCH_NUMA=1) and left alone, three times each, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05p; mkdir -p $O
lscpu | grep -i "numa\|socket\|model name" > $O/lscpu.txt 2>&1
cat /sys/bus/pci/devices/*/numa_node 2>/dev/null | sort | uniq -c > $O/pci_numa_nodes.txt
for i in 1 2 3; do
	( timeout 200 env J40_BENCH_NUMA=1 python bench.py --skip-sections --steps 10 --warmup 2 ) >> $O/bound.jsonl 2>> $O/bound.err; echo "bound$i rc=$?" >> $O/rc.txt
	( timeout 200 python bench.py --skip-sections --steps 10 --warmup 2 ) >> $O/free.jsonl 2>> $O/free.err; echo "free$i rc=$?" >> $O/rc.txt
done
( timeout 200 env J40_BENCH_NUMA=1 python bench.py --skip-sections --steps 20 --warmup 5 ) >> $O/bound.jsonl 2>> $O/bound.err; echo "bound_20 rc=$?" >> $O/rc.txt
cat $O/rc.txt; cat $O/lscpu.txt
python - <<'PY'
import json
for f in ("bound", "free"):
    for l in open("gpurun_out/r05p/%s.jsonl" % f):
        if not l.startswith("{"): continue
        r = json.loads(l); print(f, r["steps"], r["value"], r["ms_per_step"], r["pcie"]["achieved_gb_per_s"], r["pipeline"]["numa"], r["pipeline"]["host_stage_ms_per_frame"])
PY
