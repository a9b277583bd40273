"""Where the GPU's time goes in `call_mods` on a trained checkpoint (split3): runs `python -m ccsmeth_amd call_mods --io native --no_sort` on a
synthetic HiFi BAM under `rocprofv3 --kernel-trace` and reads the trace: the busy fraction of the steady state (union of kernel intervals over
the span between the first and the last GRU launch), the idle gaps by size, and the time per kernel and launch shape.
usage: python tools/call_mods_gpu_timeline.py [n_reads=16000] [extra call_mods args ...]      (writes to stdout)"""
import csv, glob, os, subprocess, sys, tempfile, time
from collections import defaultdict, OrderedDict
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ccsmeth_amd.utils import benchdata

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
extra = sys.argv[2:]
tmp = tempfile.mkdtemp(prefix="ccsm_tl_", dir=os.environ.get("TMPDIR", "/tmp"))
inp, ckpt = os.path.join(tmp, "in.bam"), os.path.join(tmp, "m.ckpt")
gen_s, nbytes = benchdata.write_synthetic_hifi_bam(inp, n_reads, 15000)
wt = dict(np.load(os.path.join(ROOT, "tests", "golden", "trained", "planted7_5000.npz")))
torch.save(OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in wt.items()), ckpt)
env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), CCSM_CALLMODS_REPORT=os.path.join(tmp, "rep.json"))
cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, "prof"), "--", sys.executable, "-m", "ccsmeth_amd", "call_mods",
       "-i", inp, "-m", ckpt, "-o", os.path.join(tmp, "out"), "--batch_size", "12288", "--no_sort"] + extra
t0 = time.time()
p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True)
print("# %d reads, %.1f MB BAM; call_mods under rocprofv3: rc %d, %.1f s wall" % (n_reads, nbytes / 1e6, p.returncode, time.time() - t0))
for ln in (p.stdout + p.stderr).splitlines():
    if "arithmetic" in ln or "sites/s" in ln:
        print("#", ln.strip()[:200])
tr = glob.glob(os.path.join(tmp, "prof", "**", "*kernel_trace.csv"), recursive=True)
rows = list(csv.DictReader(open(tr[0])))
ev = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ccsm::", "")
    grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0) // max(int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1)) or 1), 1)
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, grid))
ev.sort()
gru = [e for e in ev if e[2].startswith("gru_layer12")]
lo, hi = gru[len(gru) // 10][0], gru[-max(1, len(gru) // 20)][1]          # steady state: skip the probe / warm-up launches and the tail
ss = [e for e in ev if e[0] >= lo and e[1] <= hi]
busy, cur_s, cur_e, gaps = 0, None, None, []
for s_, e_, _, _ in ss:
    if cur_e is None or s_ > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append(s_ - cur_e)
        cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
busy += cur_e - cur_s
span = hi - lo
print("steady state: %.3f s, GPU busy (union of kernel intervals) %.1f %%, %d gaps: total %.1f ms (%.1f %%), > 20 us: %d (%.1f ms), > 200 us: %d (%.1f ms)" % (
    span / 1e9, 100.0 * busy / span, len(gaps), sum(gaps) / 1e6, 100.0 * sum(gaps) / span, sum(g > 20e3 for g in gaps), sum(g for g in gaps if g > 20e3) / 1e6,
    sum(g > 200e3 for g in gaps), sum(g for g in gaps if g > 200e3) / 1e6))
by = defaultdict(list)
for s_, e_, nm, g in ss:
    by[(nm, g)].append((e_ - s_) / 1e3)
tot = sum(sum(v) for v in by.values())
print("| kernel | workgroups | launches | mean us | share of the kernel time |\n|---|---|---|---|---|")
for (nm, g), v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("| %s | %d | %d | %.1f | %.1f %% |" % (nm[:60], g, len(v), sum(v) / len(v), 100.0 * sum(v) / tot))
print("sum of kernel durations / span = %.3f (two streams overlap)" % (tot / span))
