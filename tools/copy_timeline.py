#!/usr/bin/env python3
"""per-second summary of a rocprofv3 --kernel-trace csv: how long the copy-back blit kernels take, the gaps between them, and which
kernels run beside the slow ones. usage: python tools/copy_timeline.py <dir with *kernel_trace.csv> [out.txt]"""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
rows.sort()
t0 = rows[0][0]
copies = [(s, e) for s, e, n in rows if "copyBuffer" in n and e - s > 500000]   # the 133 MB ones
out = []
out.append("%d dispatches, %d large copies" % (len(rows), len(copies)))
sec = collections.defaultdict(list)
for (s, e) in copies:
    sec[(s - t0) // 1000000000].append((e - s) / 1e6)
prev_end = None; gaps = collections.defaultdict(float)
for (s, e) in copies:
    if prev_end is not None and s > prev_end: gaps[(s - t0) // 1000000000] += (s - prev_end) / 1e6
    prev_end = max(prev_end or 0, e)
out.append("second: copies, mean ms, max ms, idle ms between copies")
for k in sorted(sec):
    v = sec[k]; out.append("%3d: %4d  %6.2f  %6.2f  %7.1f" % (k, len(v), sum(v) / len(v), max(v), gaps.get(k, 0.0)))
slow = [(s, e) for s, e in copies if e - s > 6000000][:2000]
beside = collections.Counter()
others = [(s, e, n) for s, e, n in rows if "copyBuffer" not in n]
for (s, e) in slow[::10]:
    for (a, b, n) in others:
        if a < e and b > s: beside[n] += 1
out.append("kernels overlapping the slow copies (sampled): " + ", ".join("%s x%d" % kv for kv in beside.most_common(8)))
fast = [(s, e) for s, e in copies if e - s < 3000000]
beside = collections.Counter()
for (s, e) in fast[::20]:
    for (a, b, n) in others:
        if a < e and b > s: beside[n] += 1
out.append("kernels overlapping the fast copies (sampled): " + ", ".join("%s x%d" % kv for kv in beside.most_common(8)))
text = "\n".join(out); print(text)
if len(sys.argv) > 2: open(sys.argv[2], "w").write(text + "\n")
