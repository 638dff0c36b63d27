#!/bin/bash
# round 6, call Q: one 8K image through the public API under rocprofv3 --kernel-trace --memory-copy-trace: the two entropy launches of the
# single-image path's two phases side by side, the pixel kernels' two passes, the image's copy and the rectangles' kernel (a timeline of the last call)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06q; mkdir -p $O
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/kt_two -- python $GRAFT_REPO_ROOT/tools/latency_probe.py 4 > $O/probe.json 2> $O/probe.err ); echo "trace rc=$?" >> $O/rc.txt
python - <<'PY' > $O/timeline_single_image_two_phases.txt
import csv, glob
kt = glob.glob("/tmp/kt_two/**/*kernel_trace.csv", recursive=True)[0]
mc = glob.glob("/tmp/kt_two/**/*memory_copy_trace.csv", recursive=True)
rows = []
for r in csv.DictReader(open(kt)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel", r["Kernel_Name"].replace("j40hip::", "").split("(")[0][:60], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
for f in mc:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r.get("Direction", "") + " " + r.get("Name", "")[:40], "", ""))
rows.sort()
# the last k_hf_entropy_fast pair of the 8K frame: find launches of k_hf_entropy_fast lasting > 8 ms, take the last two and print everything from 1 ms before the first to 5 ms after the second
long_ = [i for i, r in enumerate(rows) if r[2] == "kernel" and "k_hf_entropy_fast" in r[3] and r[1] - r[0] > 8_000_000]
if len(long_) >= 2:
    # the 8K calls come first in the probe (then the 4K frame): the last pair whose durations differ clearly belongs to one call
    a, b = long_[0], long_[1]
    pairs = [(long_[k], long_[k + 1]) for k in range(len(long_) - 1) if abs(rows[long_[k]][0] - rows[long_[k + 1]][0]) < 2_000_000]
    a, b = pairs[len(pairs) // 2 - 1] if len(pairs) > 2 else pairs[-1]
    t0 = min(rows[a][0], rows[b][0]) - 1_000_000; t1 = max(rows[a][1], rows[b][1]) + 4_000_000
    print("one j40_next_frame call of an 8K image in two phases (ms from the first entropy launch's start - 1; start, duration, what, queue / stream)")
    for r in rows:
        if r[0] >= t0 and r[0] <= t1 and (r[1] - r[0] > 20_000 or r[2] == "copy"):
            print("%9.3f %9.3f  %-6s %-62s %s %s" % ((r[0] - t0) / 1e6, (r[1] - r[0]) / 1e6, r[2], r[3], r[4], r[5]))
else:
    print("no pair of long entropy launches found", len(long_))
PY
f=$(find /tmp/kt_two -name '*kernel_stats.csv' | head -1); cp "$f" $O/kernel_stats_single_image.csv 2>/dev/null
cat $O/rc.txt; cut -c1-300 $O/probe.json; head -40 $O/timeline_single_image_two_phases.txt
