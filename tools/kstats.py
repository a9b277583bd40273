#!/usr/bin/env python
"""Top rows of a rocprofv3 kernel_stats.csv with short names: python tools/kstats.py <dir-or-csv> [rows=14]"""
import csv, os, sys
p = sys.argv[1]
if os.path.isdir(p):
    p = [os.path.join(d, f) for d, _, fs in os.walk(p) for f in fs if f.endswith("kernel_stats.csv")][0]
for r in list(csv.DictReader(open(p)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%-48s calls %6s avg_us %9.1f pct %6s" % (n[:48], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
