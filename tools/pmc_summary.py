#!/usr/bin/env python
"""Mean per launch of every counter in the rocprofv3 --pmc CSV outputs under a directory tree (one sub-directory per pass).
usage: python tools/pmc_summary.py <dir> <out.md> [title]"""
import csv, glob, os, sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
title = sys.argv[3] if len(sys.argv) > 3 else "PMC summary"
rows = []
for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    pname = os.path.relpath(path, root).split(os.sep)[0]
    acc, dur, cnt = defaultdict(float), defaultdict(float), defaultdict(set)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[(k, r["Counter_Name"])] += float(r["Counter_Value"])
            cnt[k].add(r["Dispatch_Id"])
            if "Start_Timestamp" in r and r["Counter_Name"]:
                dur[(k, r["Dispatch_Id"])] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    for (k, c), v in sorted(acc.items()):
        if any(s in k for s in ("gru_layer", "attn_fc")):
            rows.append((k, "%s (%s pass)" % (c, pname), v / max(len(cnt[k]), 1)))
    for k in sorted(cnt):
        if any(s in k for s in ("gru_layer", "attn_fc")):
            d = [v for (kk, _), v in dur.items() if kk == k]
            if d:
                rows.append((k, "duration_us (%s pass)" % pname, sum(d) / len(d)))
with open(out, "w") as f:
    f.write("# %s\n\n| kernel | counter | mean per launch |\n|---|---|---|\n" % title)
    for k, c, v in rows:
        f.write("| %s | %s | %.5g |\n" % (k.replace("ccsm::", ""), c, v))
print(open(out).read())
