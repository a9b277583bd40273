#!/usr/bin/env python
"""ISA gate of the kernels whose MFMAs are asm statements (ccsm_gru_f3s.hip, ccsm_gru_mx16.hip).

hipcc pads no hazard whose producer or consumer sits inside an asm string and counts none of its memory operations, so the safety of these
kernels rests on properties of the CODE OBJECT, not of the source.  This compiles their product instantiations for gfx950 (a small
translation unit: ~1 min) and fails on
  1. scratch or spills in a GRU kernel;
  2. an MFMA whose D differs from its C (the accumulate chain is tied: a copy would sit between two asm statements, unpadded);
  3. an MFMA with a source register written by a vector-ALU instruction fewer than 2 wait states earlier (VALU write -> MFMA read);
  4. any other instruction touching a register an MFMA wrote fewer than `passes + 3` wait states earlier (XDL write -> VALU / DS / VMEM
     access; 4 passes for the 16-wide shapes -> 7 states; mfma_drain() provides 16);
  5. (mx16) a counted `s_waitcnt vmcnt(N)` in front of a pair's barrier whose N exceeds the loads the wave has issued since its part of the
     awaited transfer (the one issued RS - 1 barriers earlier): loads retire in order, stores are not counted.
Used by tests/test_isa_gate.py (also on deliberately broken builds) and by __graft_entry__.build().
    python tools/isa_gate.py [-DNAME ...] ["MX16(false, 3)" ...] [--keep] [--quiet]        exit code 0 = clean"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ccsmeth_amd", "csrc")
RING = 4            # kMxRS: ring slots of the mx16 kernels

TU = r"""
#include <hip/hip_runtime.h>
#include <type_traits>
#include <cstdint>
#include "%(c)s/ccsm_kernels.hip"
#include "%(c)s/ccsm_gru_f8.hip"
#include "%(c)s/ccsm_gru_mx.hip"
#include "%(c)s/ccsm_gru_f3.hip"
#include "%(c)s/ccsm_gru_f3s.hip"
#include "%(c)s/ccsm_gru_mx16.hip"
namespace ccsm {
#define F3S12(NB) template __global__ void gru_layer12_f3s_kernel<NB, false>(const uint4*, uint4*, const uint4*, const float*, const float*, int, unsigned long long*);
#define F3S0(NB) template __global__ void gru_layer0_f3s_kernel<NB>(const uint4*, uint4*, const uint4*, const float*, const float*, int);
#define MX16(F8, NB) template __global__ void gru_layer12_mx16_kernel<F8, NB, false>(const uint4*, uint4*, const uint4*, const float*, const float*, int, unsigned long long*);
#define MX12(F8, NB) template __global__ void gru_layer12_mx_kernel<F8, false, false, false, NB>(const uint4*, uint4*, const uint4*, const float*, const float*, int, unsigned long long*);   // (compiler-scheduled: not gated; tools/cu_issue_sim.py compiles variants of it)
%(inst)s
}
"""
# the product instantiations (what launch_run can launch)
ALL = ["F3S12(1)", "F3S12(2)", "F3S12(3)", "F3S0(1)", "F3S0(2)", "F3S0(3)",
       "MX16(false, 1)", "MX16(false, 2)", "MX16(false, 3)", "MX16(true, 1)", "MX16(true, 2)", "MX16(true, 3)"]


def compile_asm(defines, keep=False, only=None):
    d = tempfile.mkdtemp(prefix="ccsm_isa_gate_")
    src = os.path.join(d, "gate_tu.hip")
    out = os.path.join(d, "gate_tu.s")
    with open(src, "w") as f:
        f.write(TU % {"c": CSRC, "inst": " ".join(only or ALL)})
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src, "-o", out] + list(defines)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stderr[-4000:])
    s = open(out).read()
    if keep:
        print("kept:", out)
    else:
        for p in (src, out):
            os.remove(p)
        os.rmdir(d)
    return s


def regs(tok):
    tok = tok.strip()
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def all_vregs(ops):
    out = set()
    for t in ops:
        out |= regs(t)
    return out


def operands(ln):
    parts = ln.split(None, 1)
    if len(parts) < 2:
        return []
    body = parts[1]
    # operands end where the modifiers start (first token without a comma in front that is not a register / literal): split on commas, then
    # cut each piece at the first blank
    return [t.strip().split(" ")[0] for t in body.split(",")]


def mfma_passes(ln):
    if "32x32" in ln:
        return 8 if "x16_f16" in ln else 16
    return 4        # v_mfma_f32_16x16x32_f16, v_mfma_scale_f32_16x16x128_f8f6f4 on fp4 / fp6 operands


def kernel_bodies(s):
    for m in re.finditer(r"^(_Z\w+):", s, re.M):
        name = m.group(1)
        end = s.index(".Lfunc_end", m.start())
        yield name, s[m.start():end]


def scratch_bytes(s, name):
    # the metadata entry of the kernel: fields precede and follow .name; take the entry delimited by "  - .agpr_count" markers
    k = s.find(".name:           " + name + "\n")
    if k < 0:
        return None, None
    a = s.rfind("  - .agpr_count", 0, k)
    b = s.find("  - .agpr_count", k)
    ent = s[a:b if b > 0 else len(s)]
    ps = re.search(r"\.private_segment_fixed_size: (\d+)", ent)
    sp = re.search(r"\.vgpr_spill_count: (\d+)", ent)
    vg = re.search(r"\.vgpr_count: +(\d+)", ent)
    return (int(ps.group(1)) if ps else None, int(sp.group(1)) if sp else None, int(vg.group(1)) if vg else None)


class Ins(str):
    """an instruction's text; .in_asm = it comes from an asm statement of the source"""
    in_asm = False


def instructions(body):
    out = []
    in_asm = False
    for ln in body.splitlines():
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") and not t.startswith(".LBB"):
            continue
        t = t.split(";")[0].strip()
        if t:
            i = Ins(t)
            i.in_asm = in_asm
            out.append(i)
    return out


def check_hazards(name, ins, report):
    """rules 2-4 on the linear instruction sequence (control flow ignored: the loops are straight-line bodies)"""
    bad = 0
    n_mfma = tied = 0
    state = 0
    valu_write = {}         # vgpr -> state index of the last VALU write
    mfma_write = {}         # vgpr -> (state index, passes) of the last MFMA write
    for ln in ins:
        if ln.startswith(".LBB") or ln.endswith(":"):
            continue
        op = ln.split(None, 1)[0]
        mm = re.match(r"s_nop (\d+)", ln)
        if mm:
            state += int(mm.group(1)) + 1
            continue
        ops = operands(ln)
        if op.startswith("v_mfma"):
            n_mfma += 1
            d, a, b, c = ops[0], ops[1], ops[2], ops[3]
            if d == c:
                tied += 1
            else:
                bad += 1
                report("  %s: MFMA with D != C: %s" % (name[:60], ln[:100]))
            src = regs(a) | regs(b) | regs(c)
            for r in src:
                if r in valu_write and state - valu_write[r] < 2 + 1:        # the write itself occupies one state: need >= 2 states in between
                    bad += 1
                    report("  %s: MFMA reads v%d %d state(s) behind a VALU write: %s" % (name[:60], r, state - valu_write[r] - 1, ln[:100]))
                    break
            for r in regs(a) | regs(b):
                if r in mfma_write and state - mfma_write[r][0] < mfma_write[r][1] + 3 + 1:
                    bad += 1
                    report("  %s: MFMA reads v%d as A / B right behind the MFMA that wrote it: %s" % (name[:60], r, ln[:100]))
                    break
            p = mfma_passes(ln)
            for r in regs(d):
                mfma_write[r] = (state, p)
                valu_write.pop(r, None)
            state += 1
            continue
        touched = all_vregs(ops)
        for r in touched:
            if r in mfma_write and state - mfma_write[r][0] < mfma_write[r][1] + 3 + 1:
                bad += 1
                report("  %s: v%d touched %d state(s) behind the MFMA that wrote it (needs %d): %s"
                       % (name[:60], r, state - mfma_write[r][0] - 1, mfma_write[r][1] + 3, ln[:100]))
                break
        if op.startswith("v_") and not op.startswith("v_cmp") and not op.startswith("v_nop") and ops:
            for r in regs(ops[0]):
                valu_write[r] = state
                mfma_write.pop(r, None)
            if op.startswith("v_permlane") or op.startswith("v_swap"):
                for r in regs(ops[1]) if len(ops) > 1 else ():
                    valu_write[r] = state
        elif op.startswith(("ds_read", "buffer_load", "global_load", "scratch_load")) and ops and "lds" not in ln.split():
            for r in regs(ops[0]):
                valu_write.pop(r, None)
                mfma_write.pop(r, None)
        state += 1
    return bad, n_mfma, tied


def check_waits(name, ins, report):
    """rule 5 for the mx16 kernels: find the step loop (the block between the label the last back edge targets and that branch), list its
    events in program order and check every barrier's counted wait"""
    labels = {ln[:-1]: i for i, ln in enumerate(ins) if ln.startswith(".LBB") and ln.endswith(":")}
    back = None
    for i, ln in enumerate(ins):
        m = re.match(r"s_cbranch_\w+ (\.LBB\d+_\d+)", ln) or re.match(r"s_branch (\.LBB\d+_\d+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            if back is None or (i - labels[m.group(1)]) > (back[1] - back[0]):
                back = (labels[m.group(1)], i)
    if back is None:
        report("  %s: no loop found" % name[:60])
        return 1, 0
    # the rotated loop may enter in the middle: take one full iteration starting at the first barrier behind the loop head, wrapping around
    loop = ins[back[0]:back[1] + 1]
    ev = []                     # ('L', n) loads, ('T',) transfer instruction, ('W', n) counted wait, ('B',) barrier
    for ln in loop:
        if ln.startswith("buffer_load") or ln.startswith("global_load") or ln.startswith("scratch_load"):
            ev.append(("T",) if ln.split()[-1] == "lds" else ("L",))
        elif ln.startswith("s_waitcnt") and "vmcnt" in ln and ln.in_asm:       # (the compiler's own waits are not the counted ones)
            m = re.search(r"vmcnt\((\d+)\)", ln)
            ev.append(("W", int(m.group(1))))
        elif ln.startswith("s_barrier"):
            ev.append(("B",))
    nb = sum(1 for e in ev if e[0] == "B")
    if nb != 32:
        report("  %s: %d barriers in the step loop (expected 32)" % (name[:60], nb))
        return 1, 0
    # segments between barriers: seg[j] = events behind barrier j up to barrier j + 1 (cyclic); the waits in front of barrier j + 1 are at its end
    first = next(i for i, e in enumerate(ev) if e[0] == "B")
    ev = ev[first:] + ev[:first]
    segs, cur = [], []
    for e in ev[1:] + [("B",)]:
        if e[0] == "B":
            segs.append(cur)
            cur = []
        else:
            cur.append(e)
    # which barrier is consumption 0?  The step's first barrier follows the back edge: barrier index of ev[0] in program order = number of
    # barriers between the loop head and it.  All that matters here is cyclic distance, so no need to know.
    bad = 0
    exact = 0
    flat = []                                               # (segment, event) in cyclic program order
    for j in range(32):
        for e in segs[j]:
            flat.append((j, e))
    # per segment the transfer's requests: the first is `a` (every wave that moves a fragment), a second one `b` (waves 0-3 at 96 rows)
    tidx = {j: [i for i, (sj, e) in enumerate(flat) if sj == j and e[0] == "T"] for j in range(32)}
    for j in range(32):
        seg = segs[j]                                       # ends with the waits of barrier j + 1
        waits = []
        k = len(seg)
        while k > 0 and seg[k - 1][0] == "W":
            waits.append(seg[k - 1][1])
            k -= 1
        if not waits:
            report("  %s: barrier without a counted wait" % name[:60])
            bad += 1
            continue
        end = max(i for i, (sj, e) in enumerate(flat) if sj == j) - len(waits)      # index of the last event in front of the waits
        src = (j - (RING - 2)) % 32
        if not tidx[src] or len(tidx[src]) > 2:
            report("  %s: %d transfer requests behind barrier %d" % (name[:60], len(tidx[src]), src))
            bad += 1
            continue

        def younger(start, two_request_wave):
            """requests a wave issues behind flat[start] up to the wait: loads, `a` of later transfers, `b` only for a two-request wave"""
            n = 0
            i = start
            while True:
                i = (i + 1) % len(flat)
                sj, e = flat[i]
                if e[0] == "L":
                    n += 1
                elif e[0] == "T":
                    is_b = len(tidx[sj]) == 2 and i == tidx[sj][1]
                    if two_request_wave or not is_b:
                        n += 1
                if i == end:
                    return n
        cnt_lo = younger(tidx[src][0], False)
        cnt_hi = younger(tidx[src][-1], True)
        cap = lambda c: min(c, 63)
        # two variants of the wait (one per wave class) when the transfer has two requests; which is which cannot be read off the code object
        # without following its scalar branch: the pair must match the two counts in one of the two assignments, and since either could then be
        # taken by either class BOTH must be safe for the class that allows fewer... no: each variant is guarded by the wave class in the source
        # (xfer_wait), so the check is that the SET of immediates equals what the two classes may use - never more than the class's count
        if len(waits) == 1:
            lo = hi = waits[0]
            ok = lo <= cap(min(cnt_lo, cnt_hi)) if len(tidx[src]) == 2 else lo <= cap(cnt_lo)
        else:
            a_, b_ = waits[0], waits[1]
            ok = (a_ <= cap(cnt_lo) and b_ <= cap(cnt_hi)) or (b_ <= cap(cnt_lo) and a_ <= cap(cnt_hi))
            # (an assignment that is safe one way round only because the immediates happen to be small passes too: the counts are what is exact below)
            lo, hi = (a_, b_) if (a_ == cap(cnt_lo) and b_ == cap(cnt_hi)) else (b_, a_)
        if not ok:
            bad += 1
            report("  %s: wait(s) %s in front of barrier %d, but a one-request wave has issued only %d and a two-request wave %d requests behind its part of the awaited transfer"
                   % (name[:60], waits, (j + 1) % 32, cnt_lo, cnt_hi))
        elif lo == cap(cnt_lo) and hi == cap(cnt_hi if len(waits) > 1 else cnt_lo):
            exact += 1
    return bad, exact


def shipped_kernels(lib=None):
    """The kernels of a BUILT libccsm.so, read from the gfx950 code object inside it (no recompilation; seconds):
    -> {kernel name: dict(vgpr_count, private_segment_fixed_size, vgpr_spill_count, sgpr_spill_count, group_segment_fixed_size)}"""
    import tempfile
    lib = lib or os.path.join(ROOT, "ccsmeth_amd", "lib", "libccsm.so")
    llvm = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin")
    with tempfile.TemporaryDirectory(prefix="ccsm_co_") as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.run([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(d, "unused.so")], check=True)
        subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        "--input=" + fat, "--output=" + co], check=True)
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    out, cur = {}, None
    for ln in notes.split("\n"):
        m = re.match(r"^  (- | {2})\.(\w+):\s+(\S+)", ln)        # the keys of a kernel's own entry (its arguments sit deeper)
        if not m:
            continue
        if m.group(1) == "- ":
            cur = {}
        k, v = m.group(2), m.group(3)
        if cur is None:
            continue
        if k in ("vgpr_count", "private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size"):
            cur[k] = int(v)
        elif k == "name":
            out[v] = cur
    return out


def run(defines=(), quiet=False, keep=False, only=None):
    """-> (number of findings, report lines); only = a subset of ALL (the instantiation macros of the translation unit)"""
    s = compile_asm(defines, keep, only)
    lines = []
    report = lines.append
    total = 0
    for name, body in kernel_bodies(s):
        if "f3s_kernel" not in name and "mx16_kernel" not in name:
            continue
        ins = instructions(body)
        ps, sp, vg = scratch_bytes(s, name)
        bad = 0
        if ps or sp:
            bad += 1
            report("  %s: scratch %s B, %s spills" % (name[:60], ps, sp))
        if any(ln.startswith("scratch_") for ln in ins):
            bad += 1
            report("  %s: scratch instructions in the body" % name[:60])
        hb, n, tied = check_hazards(name, ins, report)
        bad += hb
        extra = ""
        if "mx16_kernel" in name:
            wb, exact = check_waits(name, ins, report)
            bad += wb
            extra = ", 32 counted waits checked (%d exact)" % exact
        report("%-74s %3s VGPRs, scratch %s, %5d MFMAs (%d tied)%s: %s" % (name[:74], vg, ps, n, tied, extra, "FAIL" if bad else "ok"))
        total += bad
    if not quiet:
        print("\n".join(lines))
    return total, lines


if __name__ == "__main__":
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    only = [a for a in sys.argv[1:] if a in ALL] or None
    bad, _ = run(defs, quiet="--quiet" in sys.argv, keep="--keep" in sys.argv, only=only)
    sys.exit(1 if bad else 0)
