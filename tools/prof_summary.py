#!/usr/bin/env python3
"""exports the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) as text"""
import glob
import sqlite3
import sys


def main(path, out):
    dbs = glob.glob(path + "/**/*.db", recursive=True) + glob.glob(path + "/*.db")
    con = sqlite3.connect(dbs[0])
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    spread = {}
    try:   # (shortest / longest launch per kernel: the shortest is the launch that had the device most to itself)
        for name, lo, hi in con.execute("select name, min(duration), max(duration) from kernels group by name").fetchall():
            spread[name] = (lo / 1e3, hi / 1e3)
    except sqlite3.Error:
        pass
    with open(out, "w") as fp:
        fp.write("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)\n")
        fp.write("%-110s %8s %14s %12s %8s %12s %12s\n" % ("kernel", "calls", "total_us", "avg_us", "pct", "min_us", "max_us"))
        for name, calls, total, avg, pct in rows:
            lo, hi = spread.get(name, (float("nan"), float("nan")))
            fp.write("%-110s %8d %14.1f %12.2f %8.3f %12.2f %12.2f\n" % (name[:110], calls, total, avg, pct, lo, hi))
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
