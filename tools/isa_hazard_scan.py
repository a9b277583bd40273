#!/usr/bin/env python
"""Hazard audit of the kernels whose MFMAs are asm statements (ccsm_gru_f3s.hip): hipcc pads no hazard whose consumer is inside an asm
string, so this lists every v_mfma of those kernels whose A / B / C registers were written by a vector-ALU instruction fewer than two
wait states earlier (what a VALU write -> MFMA read wants), and the share of MFMAs with D == C (the tied accumulate chain, which needs none).
    python tools/isa_hazard_scan.py [kernel-name filter = f3s]       (compiles ccsm_api.hip to assembly: ~3 min)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flt = sys.argv[1] if len(sys.argv) > 1 else "f3s"
out = os.path.join(tempfile.mkdtemp(prefix="ccsm_haz_"), "k.s")
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", os.path.join(ROOT, "ccsmeth_amd", "csrc", "ccsm_api.hip"), "-o", out],
               check=True, stderr=subprocess.DEVNULL)
s = open(out).read()


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


bad = 0
for m in re.finditer(r"^(_Z\w+):", s, re.M):
    name = m.group(1)
    if flt not in name:
        continue
    body = [ln.strip() for ln in s[m.start():s.index(".Lfunc_end", m.start())].splitlines()
            if ln.strip() and not ln.strip().startswith(";") and not ln.strip().startswith(".")]
    n = tied = short = 0
    for k, ln in enumerate(body):
        if not ln.startswith("v_mfma"):
            continue
        n += 1
        ops = [t.strip() for t in ln.split(None, 1)[1].split(",")]
        tied += ops[0] == ops[3]
        src = set().union(*[regs(t) for t in ops[1:4]])
        for back in range(1, 3):
            p = body[k - back]
            if p.startswith("v_") and not p.startswith("v_mfma") and not p.startswith("v_cmp"):
                if regs(p.split(None, 1)[1].split(",")[0].strip()) & src:
                    ws = 0
                    for q in body[k - back + 1:k]:
                        mm = re.match(r"s_nop (\d+)", q)
                        ws += (int(mm.group(1)) + 1) if mm else 1
                    if ws < 2:
                        short += 1
                        print("  %s\n     <- %s (%d state(s) between)" % (ln[:90], p[:70], ws))
                    break
    bad += short
    print("%-70s %5d MFMAs, %5d with D == C, %d behind a VALU write of a source by fewer than 2 states" % (name[:70], n, tied, short))
sys.exit(1 if bad else 0)
