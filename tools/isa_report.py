#!/usr/bin/env python
"""ISA report of the gfx950 code object of libccsm: per kernel the register / scratch figures of the metadata and counts of
the instructions that matter for the prefetch pipeline (full vmcnt drains, flat loads, scratch traffic, MFMAs).

    python tools/isa_report.py [--filter gru_layer_f8] [--out profiles/r02_isa.md]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_s(src, extra):
    out = os.path.join(tempfile.mkdtemp(prefix="ccsm_isa_"), "k.s")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src, "-o", out] + extra
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


def report(spath, flt):
    text = open(spath).read().split("\n")
    bodies, cur = {}, None
    for ln in text:
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
        if m:
            cur = m.group(1)
            bodies[cur] = []
            continue
        if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
            cur = None
        if cur is not None:
            bodies[cur].append(ln)
    meta = {}
    name = None
    for ln in text:
        m = re.match(r"\s+\.name:\s+(\S+)", ln)
        if m:
            name = m.group(1)
            meta.setdefault(name, {})
        for key in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count",
                    "group_segment_fixed_size"):
            m = re.match(r"\s+\.%s:\s+(\d+)" % key, ln)
            if m and name:
                meta[name][key] = int(m.group(1))
    rows = []
    for k, body in bodies.items():
        if k not in meta or (flt and flt not in k):
            continue
        ins = [b.strip() for b in body if b.startswith("\t") and not b.strip().startswith((".", ";"))]
        cnt = lambda pat: sum(1 for i in ins if re.search(pat, i))
        rows.append((k, meta[k], {"instructions": len(ins), "vmcnt(0)": cnt(r"s_waitcnt.*vmcnt\(0\)"), "s_waitcnt vmcnt": cnt(r"s_waitcnt.*vmcnt"),
                                  "flat_load": cnt(r"^flat_load"), "scratch_": cnt(r"^scratch_"), "v_mfma": cnt(r"^v_mfma"),
                                  "buffer_load": cnt(r"^buffer_load"), "global_load": cnt(r"^global_load"), "ds_read": cnt(r"^ds_read"),
                                  "s_barrier": cnt(r"^s_barrier")}))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--src", default=os.path.join(ROOT, "ccsmeth_amd", "csrc", "ccsm_api.hip"))
    ap.add_argument("--filter", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("extra", nargs="*")
    a = ap.parse_args()
    rows = report(compile_s(a.src, a.extra), a.filter)
    lines = ["| kernel | vgpr | sgpr | scratch B | vgpr spills | LDS B | instr | v_mfma | vmcnt(0) | vmcnt waits | flat_load | scratch_* | buffer_load | global_load | ds_read | s_barrier |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for k, m, c in rows:
        dem = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
        lines.append("| `%s` | %d | %d | %d | %d | %d | %d | %d | %d | %d | %d | %d | %d | %d | %d | %d |" % (
            dem, m.get("vgpr_count", -1), m.get("sgpr_count", -1), m.get("private_segment_fixed_size", -1), m.get("vgpr_spill_count", -1),
            m.get("group_segment_fixed_size", -1), c["instructions"], c["v_mfma"], c["vmcnt(0)"], c["s_waitcnt vmcnt"], c["flat_load"],
            c["scratch_"], c["buffer_load"], c["global_load"], c["ds_read"], c["s_barrier"]))
    txt = "\n".join(lines)
    print(txt)
    if a.out:
        with open(a.out, "w") as f:
            f.write("# ISA report (hipcc --offload-arch=gfx950 -O3, tools/isa_report.py)\n\n" + txt + "\n")


if __name__ == "__main__":
    main()
