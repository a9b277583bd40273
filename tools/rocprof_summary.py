#!/usr/bin/env python
"""Turn a rocprofv3 rocpd sqlite result (--kernel-trace --stats) into a small CSV of per-kernel stats.
usage: python tools/rocprof_summary.py <results.db> <out.csv>"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    for name, calls, tot, avg, pct in rows:
        if pct < 0.001:
            continue
        w.writerow([name.split("(")[0].replace("void ", ""), calls, "%.3f" % tot, "%.3f" % avg, "%.4f" % pct])
print(open(sys.argv[2]).read())
