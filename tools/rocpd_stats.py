#!/usr/bin/env python3
"""per-kernel summary (calls, total, average, min, max, share) of a rocprofv3 run that wrote a rocpd database (results.db) instead of CSV.
usage: python tools/rocpd_stats.py path/to/NNN_results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
disp = next(t for t in tables if t.startswith("rocpd_kernel_dispatch")); sym = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
rows = list(db.execute("select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3 "
                       "from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by 3 desc" % (disp, sym)))
tot = sum(r[2] for r in rows)
print("%-100s %6s %10s %10s %10s %10s %6s" % ("kernel", "calls", "total ms", "avg us", "min us", "max us", "%"))
for r in rows:
    print("%-100s %6d %10.2f %10.1f %10.1f %10.1f %6.1f" % (r[0][:100], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
