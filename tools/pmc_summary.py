#!/usr/bin/env python3
"""per-kernel averages of rocprofv3 --pmc counters (csv output): python tools/pmc_summary.py <dir> [out.txt]"""
import csv, glob, sys, collections

def main(path, out=None):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = []
    for k, cs in sorted(rows.items()):
        lines.append(k)
        for c, v in sorted(cs.items()):
            lines.append("    %-28s n=%-4d avg=%.4g" % (c, len(v), sum(v) / len(v)))
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")

if __name__ == "__main__":
    main(*sys.argv[1:3])
