#!/usr/bin/env python3
"""exports the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) as text"""
import glob
import sqlite3
import sys


def main(path, out):
    dbs = glob.glob(path + "/**/*.db", recursive=True) + glob.glob(path + "/*.db")
    con = sqlite3.connect(dbs[0])
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out, "w") as fp:
        fp.write("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)\n")
        fp.write("%-110s %8s %14s %12s %8s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for name, calls, total, avg, pct in rows:
            fp.write("%-110s %8d %14.1f %12.2f %8.3f\n" % (name[:110], calls, total, avg, pct))
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
