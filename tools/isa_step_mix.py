#!/usr/bin/env python
"""What ONE STEP of a GRU kernel asks of each unit of a CU, counted on the code object, next to what the counters measured.

The step loop of a GRU kernel is straight-line code (its inner loops are unrolled), so the instructions between the loop's label and its backward
branch are what a wave executes per step; the only wave-dependent part is the x transfers' extra fragment (waves 0-3 of 8: counted at 0.5, see
step_loop).  From them, per step, averaged over the eight waves:

  matrix pipe   cycles per SIMD = 2 waves x sum over MFMAs of (passes x 4)       [32x32x16 f16 and 32x32x64 fp4 x fp6: 8 passes; 16x16x32 f16 and
                                                                                  16x16x128 fp4 x fp6: 4 passes - what SQ_VALU_MFMA_BUSY_CYCLES counts]
  vector memory cycles per CU   = 8 waves x sum over requests of bytes / 64      [the L1 path moves 64 B/clk: a 1-KiB request (dwordx4 per lane, to
                                                                                  registers or to LDS) holds it 16 cycles; DESIGN 7.6]
  LDS reads     cycles per CU   = 8 waves x (4 per ds_read_b128, 2 per ds_read_b64, 8 per ds_read2_b64)   [MI355X_MICROARCH, LDS table]
  L2 requests   per launch      = bytes requested / 128 B                         [what TCC_REQ_sum counts]

and beside them the launch's counters (tools/pmc_summary.py tables under profiles/): GRBM_GUI_ACTIVE / 8 XCDs = cycles per launch,
SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs, TCC_REQ_sum.  The matrix-pipe and L2-request columns must AGREE with the counters (they do, to
the digit / within 3 %: the prologue's requests are not in the loop): that pins the counting; the interesting column is the launch's cycles against the SUM and the MAXIMUM of the first two.

    python tools/isa_step_mix.py [--asm api.s] [--pmc profiles/r06_p_pmc.md:2 profiles/r05_w_pmc_split3.md:1] [--out profiles/r06_q_step_mix.md]
    (--pmc file:rounds - rounds = workgroups of the profiled launch / 256 CUs; without --asm ccsm_api.hip is compiled to assembly, ~4 min)
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 21                                  # kSeqLen
MFMA_PASSES = {"v_mfma_f32_32x32x16_f16": 8, "v_mfma_f32_16x16x32_f16": 4, "v_mfma_scale_f32_32x32x64_f8f6f4": 8, "v_mfma_scale_f32_16x16x128_f8f6f4": 4}
REQ_BYTES = {"dwordx4": 1024, "dwordx3": 768, "dwordx2": 512, "dword": 256, "short": 128, "ubyte": 64}
LDS_CYCLES = {"ds_read_b128": 4, "ds_read_b64": 2, "ds_read_b32": 2, "ds_read2_b64": 8, "ds_read2st64_b64": 8, "ds_read2_b32": 4}
WANTED = ("gru_layer12_mx_kernel", "gru_layer0_mx_kernel", "gru_layer12_f3s_kernel", "gru_layer0_f3s_kernel", "gru_layer12_mx16_kernel")


def kernel_bodies(s):
    for m in re.finditer(r"^(_Z\w+):", s, re.M):
        yield m.group(1), s[m.start():s.index(".Lfunc_end", m.start())]


def step_loop(body):
    """[(instruction, weight)] of the step loop: from the target of the kernel's longest backward branch to that branch.  Weight 1, except
    behind a forward branch on a SCALAR condition (s_cbranch_scc* / vcc*: wave-uniform - in these kernels the x transfers' extra fragment, which
    waves 0-3 of the eight move and waves 4-7 skip, in one or both arms of the branch): 0.5.  (s_cbranch_exec* only skips code no lane wants.)"""
    lines = body.split("\n")
    labs = {m.group(1): i for i, ln in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", ln)] if m}
    best = None
    for i, ln in enumerate(lines):
        m = re.search(r"\b(?:s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", ln)
        if m and labs.get(m.group(1), i) < i and (best is None or labs[m.group(1)] <= best[0]):
            best = (labs[m.group(1)], i)
    if best is None:
        return []
    a, e = best
    weight = [1.0] * len(lines)
    for i in range(a, e):
        m = re.search(r"\bs_cbranch_(?:scc|vcc)\w*\s+(\.LBB\d+_\d+)", lines[i])
        if m and i < labs[m.group(1)] <= e:
            for j in range(i + 1, labs[m.group(1)]):
                weight[j] = 0.5
    out = []
    for j in range(a, e + 1):
        t = lines[j].strip().split(";")[0].strip()
        if lines[j].startswith("\t") and t and not t.startswith("."):
            out.append((t, weight[j]))
    return out


def mix(ins):
    c = collections.Counter()
    mfma_cyc = req_bytes = lds_cyc = 0
    for i, w in ins:
        op = i.split()[0]
        if op.startswith("v_mfma"):
            c[op] += w
            mfma_cyc += 4 * MFMA_PASSES[op] * w
        elif op.startswith(("buffer_load_", "global_load_", "buffer_store_", "global_store_")):
            width = op.split("_", 2)[2]
            c[op + (" lds" if re.search(r"\blds\b", i) else "")] += w
            req_bytes += REQ_BYTES[width] * w
        elif op.startswith("ds_read"):
            c[op] += w
            lds_cyc += LDS_CYCLES[op] * w
        elif op.startswith("ds_write"):
            c[op] += w
        elif op in ("s_barrier", "v_exp_f32_e32", "v_rcp_f32_e32"):
            c[op] += w
        elif op.startswith("v_"):
            c["other vector ALU"] += w
    return c, mfma_cyc, req_bytes, lds_cyc


def pmc_table(path):
    """{kernel (as the table prints it): {counter: value}} of a tools/pmc_summary.py table"""
    out = collections.defaultdict(dict)
    for ln in open(path):
        f = [x.strip() for x in ln.strip().strip("|").split("|")]
        if len(f) == 3 and f[1] not in ("counter", "---"):
            try:
                out[f[0]][f[1].split(" (")[0]] = float(f[2])
            except ValueError:
                pass
    return out


def short(demangled):
    m = re.match(r"(?:void )?(?:ccsm::)?(\w+(?:<[^(]*>)?)\(", demangled)
    return m.group(1) if m else demangled


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm")
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("--out")
    ap.add_argument("--all", action="store_true", help="every instantiation (default: those with counters, and the 96-row forms of the mx16 kernel)")
    a = ap.parse_args()
    asm = a.asm
    if not asm:
        asm = os.path.join(tempfile.mkdtemp(prefix="ccsm_mix_"), "api.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                        os.path.join(ROOT, "ccsmeth_amd", "csrc", "ccsm_api.hip"), "-o", asm], check=True, stderr=subprocess.DEVNULL)
    s = open(asm).read()
    pmc = {}
    for spec in a.pmc:
        path, rounds = spec.rsplit(":", 1)
        for k, v in pmc_table(path).items():
            pmc.setdefault(k, (v, int(rounds), os.path.basename(path)))
    cxxfilt = "c++filt"
    rows, detail = [], []
    for name, body in kernel_bodies(s):
        if not any(w in name for w in WANTED):
            continue
        nm = short(subprocess.run([cxxfilt, name], capture_output=True, text=True).stdout.strip())
        ins = step_loop(body)
        c, mfma_cyc, req_bytes, lds_cyc = mix(ins)
        if not mfma_cyc:
            continue
        pipe, path, lds = int(2 * mfma_cyc), int(8 * req_bytes / 64), int(8 * lds_cyc)
        row = [nm, "%g" % sum(v for k, v in c.items() if k.startswith("v_mfma")), "%g" % sum(v for k, v in c.items() if "load" in k or "store" in k),
               pipe, path, lds, pipe + path, max(pipe, path)]
        meas = pmc.get(nm)
        if meas:
            v, rounds, src = meas
            steps = STEPS * rounds
            cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8 / steps
            busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / steps
            l2_static = req_bytes * 8 * steps * 256 / 128
            row += ["%.0f" % cyc, "%.0f" % busy, "%.2f" % (cyc / (pipe + path)), "%.2f" % (cyc / max(pipe, path)),
                    "%.3g / %.3g" % (l2_static, v["TCC_REQ_sum"]) if "TCC_REQ_sum" in v else "-", src]
        else:
            if not a.all and not ("mx16" in nm and ", 3," in nm):
                continue
            row += ["-"] * 6
        rows.append(row)
        detail.append((nm, len(ins), c))
    rows.sort(key=lambda r: r[0])
    out = ["# What one step asks of a CU's units, counted on the code object, against the counters (tools/isa_step_mix.py)", "",
           "Per STEP (one of 21 per workgroup; a CU runs one 8-wave workgroup at a time, two waves per SIMD).  matrix pipe = cycles per SIMD the two",
           "waves' MFMAs occupy it; L1 path = cycles per CU the eight waves' vector-memory requests occupy it at 64 B/clk; LDS reads likewise at the",
           "guide's rates.  `measured` = GRBM_GUI_ACTIVE / 8 / steps of the profiled launch; `pipe busy` = SQ_VALU_MFMA_BUSY_CYCLES / 1024 / steps (must equal",
           "the static matrix-pipe column); `L2 requests` = static bytes / 128 B against TCC_REQ_sum per launch.", "",
           "| kernel | MFMAs / wave | requests / wave | matrix pipe | L1 path | LDS reads | sum | max | measured | pipe busy (PMC) | measured / sum | measured / max | L2 requests static / PMC | counters from |",
           "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        out.append("| `%s` | %s |" % (r[0], " | ".join(str(x) for x in r[1:])))
    out += ["", "## Instructions of the step loop, per wave", ""]
    for nm, n, c in sorted(detail):
        out.append("* `%s` (%d instructions): %s" % (nm, n, ", ".join("%g %s" % (v, k) for k, v in sorted(c.items(), key=lambda x: -x[1]))))
    text = "\n".join(out) + "\n"
    if a.out:
        open(a.out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
