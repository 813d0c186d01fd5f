#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average duration, share) of a rocprofv3 --kernel-trace run stored as a rocpd
SQLite database -> CSV on stdout (the same columns rocprofv3's kernel_stats.csv has).

    python tools/rocpd_stats.py gpurun_out/prof/r_results.db > profiles/rNN_kernel_stats.csv
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                       "from kernels group by name order by sum(duration) desc"))
total = sum(r[2] for r in rows) or 1
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
for name, calls, tot, avg, mn, mx in rows:
    print(f'"{name}",{calls},{tot},{avg:.1f},{100.0 * tot / total:.2f},{mn},{mx}')
