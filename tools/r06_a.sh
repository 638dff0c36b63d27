#!/bin/bash
# round 6, call A: why the contract clock has slow runs. (1) tools/ubench/copy_probe: 133 MB device-to-host copies issued eleven ways, idle and
# beside kernels that hold every wavefront slot; which of them are blit kernels and which SDMA transfers (rocprofv3 kernel + memory-copy
# trace per way). (2) the timed pipeline alone, alternating the copy stream's kind: default / a CU-mask stream (a hardware queue of its
# own) / two CU-mask streams / two default streams. (3) one default run with the runtime's queue log, one under rocprofv3 with the
# memory-copy trace.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06a; mkdir -p $O
( timeout 120 build/copy_probe 48 10 ) > $O/copy_probe.txt 2>&1; echo "copy_probe rc=$?" >> $O/rc.txt
# which ways are kernels, which are SDMA: one way per profiled run, beside the hogs only, 8 copies
for k in 0 2 7 8 10; do
	( cd /tmp && timeout 120 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/prof_way$k -o way$k -- $GRAFT_REPO_ROOT/build/copy_probe 8 3 $((1 << k)) 1 ) > $O/prof_way$k.txt 2>&1
	echo "prof way $k rc=$?" >> $O/rc.txt
done
python - <<'PY' > $O/prof_ways_summary.txt 2>&1
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06a"
for k in (0, 2, 7, 8, 10):
    kern = [r for f in glob.glob(O + "/prof_way%d/**/*kernel_trace.csv" % k, recursive=True) for r in csv.DictReader(open(f))]
    cop = [r for f in glob.glob(O + "/prof_way%d/**/*memory_copy_trace.csv" % k, recursive=True) for r in csv.DictReader(open(f))]
    names = {}
    for r in kern: names[r["Kernel_Name"][:50]] = names.get(r["Kernel_Name"][:50], 0) + 1
    big = [r for r in cop if int(r.get("End_Timestamp", 0)) - int(r.get("Start_Timestamp", 0)) > 500000]
    print("way", k, "kernels:", names, "| memory-copy records:", len(cop), "of which > 0.5 ms:", len(big), "|", (cop[0] if cop else None))
PY
run() {   # name, env...
	name=$1; shift
	( timeout 150 env "$@" python bench.py --skip-sections --no-cpu-baseline --steps 8 --warmup 2 ) >> $O/$name.jsonl 2>> $O/$name.err; echo "$name rc=$?" >> $O/rc.txt
}
for i in 1 2 3; do
	run default J40HIP_X=0
	run mask J40HIP_COPY_STREAM=mask
	run mask2 J40HIP_COPY_STREAM=mask J40HIP_COPY_STREAMS=2
	run two J40HIP_COPY_STREAMS=2
done
( timeout 150 env AMD_LOG_LEVEL=4 AMD_LOG_MASK=0x10 python bench.py --skip-sections --no-cpu-baseline --steps 4 --warmup 1 ) > $O/queuelog.jsonl 2> $O/queuelog.err; echo "queuelog rc=$?" >> $O/rc.txt
grep -i "queue" $O/queuelog.err | grep -v "^\[j40hip" | cut -c1-200 | head -150 > $O/queuelog_head.txt; wc -l $O/queuelog.err >> $O/queuelog_head.txt; rm -f $O/queuelog.err
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --skip-sections --no-cpu-baseline --steps 4 --warmup 1 ) > $O/prof_bench.txt 2>&1; echo "prof_bench rc=$?" >> $O/rc.txt
python tools/copy_timeline.py $O/prof_bench $O/prof_bench_copy_timeline.txt > /dev/null 2>&1
python - <<'PY' > $O/prof_bench_copies.txt 2>&1
import csv, glob, os, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06a"
cop = [r for f in glob.glob(O + "/prof_bench/**/*memory_copy_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
print(len(cop), "memory-copy records; columns:", list(cop[0].keys()) if cop else None)
by = collections.Counter((r.get("Direction"), ) for r in cop)
print(by)
big = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in cop if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 500000]
if big: print("copies > 0.5 ms:", len(big), "mean ms", sum(big) / len(big) / 1e6, "min", min(big) / 1e6, "max", max(big) / 1e6)
PY
# keep what is small
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete; find $O -name "*.rocpd" -delete
cat $O/rc.txt
cat $O/copy_probe.txt
cat $O/prof_ways_summary.txt
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06a"
for f in ("default", "mask", "mask2", "two", "queuelog"):
    try:
        for l in open("%s/%s.jsonl" % (O, f)):
            if not l.startswith("{"): continue
            r = json.loads(l); print(f, r["value"], r["ms_per_step"], r["pcie"]["achieved_gb_per_s"], r["pipeline"]["cgroup_cpu_in_region"], r["pipeline"]["host_stage_ms_per_frame"])
    except OSError as e: print(f, e)
PY
cat $O/prof_bench_copies.txt; cat $O/prof_bench_copy_timeline.txt; head -60 $O/queuelog_head.txt
