#!/usr/bin/env python
"""Per-kernel launch durations of a rocprofv3 --kernel-trace CSV, separated BY LAUNCH SHAPE (grid size): the --stats average of
bench.py's command mixes the full groups (6 x 2048 sites), the ragged last group (2 x 2048), the warm-up's first launches and the
cpu_baseline probe; bench.py's `roofline.launch_ms` is the mean over the timed region's FULL groups only.  The "timed region" rows
below are the last `full_groups` launches of the full-group shape.
usage: python tools/kernel_trace_by_shape.py <kernel_trace.csv> <out.md> [full_groups_in_timed_region=3]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
nlast = int(sys.argv[3]) if len(sys.argv) > 3 else 3
by = defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
    wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1)) or 1)
    by[(name, grid // max(wg, 1))].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
out = ["| kernel | workgroups | launches | mean us | min us | max us | mean of the last %d (timed region) us |" % nlast, "|---|---|---|---|---|---|---|"]
for (name, wgs), v in sorted(by.items(), key=lambda kv: -sum(d for _, d in kv[1])):
    v.sort()
    d = [x for _, x in v]
    if sum(d) < 50:
        continue
    out.append("| %s | %d | %d | %.1f | %.1f | %.1f | %.1f |" % (name, wgs, len(d), sum(d) / len(d), min(d), max(d), sum(d[-nlast:]) / len(d[-nlast:])))
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print("\n".join(out))
