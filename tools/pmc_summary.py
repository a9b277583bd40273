#!/usr/bin/env python
"""Mean per launch of every counter in the rocprofv3 --pmc CSV outputs under a directory tree (one sub-directory per pass).
usage: python tools/pmc_summary.py <dir> <out.md> [title] [--emit <traffic.json> --precision P --sites N --source <text>]
--emit merges into <traffic.json> (what bench.py's roofline.traffic reads) one entry per GRU layer-1/2 kernel found: bytes per site of ONE launch
= (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 / N sites (FETCH_SIZE doubled: MI355X_MICROARCH.md's gfx950 note), mean over the two layers."""
import csv, glob, os, sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
title = sys.argv[3] if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else "PMC summary"


def opt(name, default=None):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default
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

if "--emit" in sys.argv:
    import json, re
    path, prec, sites = opt("--emit"), int(opt("--precision", "4")), int(opt("--sites", "12288"))
    fetch, write = defaultdict(list), defaultdict(list)
    for k, c, v in rows:
        base = re.sub(r"<.*", "", k.replace("ccsm::", ""))
        if not base.startswith("gru_layer12"):
            continue
        if c.startswith("FETCH_SIZE"):
            fetch[base].append(v)
        elif c.startswith("WRITE_SIZE"):
            write[base].append(v)
    try:
        doc = json.load(open(path))
    except (OSError, ValueError):
        doc = {"kernels": []}
    doc["source"] = opt("--source", out)
    for base in sorted(fetch):
        if base not in write:
            continue
        f, w = sum(fetch[base]) / len(fetch[base]), sum(write[base]) / len(write[base])
        ent = {"precision": prec, "kernel": base, "sites_per_launch": sites, "fetch_size_KiB": f, "write_size_KiB": w, "launch_variants_averaged": len(fetch[base]),
               "bytes_per_site": (2.0 * f + w) * 1024.0 / sites, "formula": "(2 x FETCH_SIZE + WRITE_SIZE) KiB per launch, mean of layers 1 and 2"}
        doc["kernels"] = [e for e in doc["kernels"] if not (e.get("precision") == prec and e.get("kernel") == base)] + [ent]
    json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
    print("emitted", path)
