"""Is the GRU kernel clock/power-bound?  Runs coalesced groups for a few seconds per variant while a thread samples rocm-smi
(power, sclk), and reports per-kernel times next to the sampled clock and power.  Variants: normal synthetic weights and data;
all-zero weights (same instruction stream, no operand toggling: what DVFS gives back shows as a shorter kernel at the same
cycle count).   python tools/gpu_power.py [seconds_per_variant]"""
import json, os, subprocess, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.utils import synth

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
n, g, dev = 2048, 6, torch.device("cuda:0")
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out)
            card = d[sorted(d)[0]]
            samples.append((time.perf_counter(), {k: v for k, v in card.items() if "ower" in k or "sclk" in k or "mclk" in k}))
        except Exception as e:   # noqa: BLE001
            samples.append((time.perf_counter(), {"error": str(e)}))
        time.sleep(0.05)


th = threading.Thread(target=sampler, daemon=True)
th.start()
s = synth.synth_sites(n * g, 8)
batches = []
for b in range(g):
    sl = slice(b * n, (b + 1) * n)
    batches.append(tuple(torch.from_numpy(np.ascontiguousarray(s[k][sl])).to(dev) for k in
                         ("kmer1", "ipd1", "pw1", "npass1", "kmer2", "ipd2", "pw2", "npass2")))
for name in os.environ.get("MODES", "normal,zero_weights,normal").split(","):
    w = synth.synth_weights(7)
    if name == "zero_weights":
        w = {k: np.zeros_like(v) for k, v in w.items()}
    dm = DeviceModel(w, 0, precision=int(os.environ.get("PREC", "4")))
    ws = dm.workspace(n * g)
    ws.set_timing(True)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(8):
            for b in range(g):
                ws.group_add_torch(*batches[b], seed=1, offset=reps * n * g + b * n, h0=None if name == "normal" else "zero")
            ws.group_run()
            reps += 1
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    ms, nr = ws.timing_mean()
    mine = [v for (ts, v) in samples if t0 + 0.5 < ts < t1]
    print("%-13s %d groups in %.2f s = %.3g sites/s; kernel ms [gru0 gru1 gru2 attn fin] %s (n=%d)" % (name, reps, t1 - t0, reps * n * g / (t1 - t0), np.round(ms, 4), nr))
    def num(d, key):
        for k, v in d.items():
            if key in k:
                try:
                    return float(str(v).strip("()").replace("Mhz", ""))
                except ValueError:
                    return float("nan")
        return float("nan")
    pw = [num(m, "ower") for m in mine]
    ck = [num(m, "sclk clock speed") for m in mine]
    print("   rocm-smi (%d samples): power W median %.0f, sclk MHz median %.0f" % (len(mine), float(np.median(pw)) if pw else -1, float(np.median(ck)) if ck else -1))
    ws.close(); dm.close()
stop = True
