"""Runs tools/ubench/_build/mfma_power_ceiling on the GPU while sampling rocm-smi (sclk, package power) and prints, per mode, the
sustained fp16-MFMA rate next to the clock and power it was measured at: the power-capped ceiling bench.py's
`roofline.peak_power_capped` quotes.   python tools/ubench/run_power_ceiling.py [seconds per mode] > profiles/r03_a_power_ceiling.log"""
import json
import os
import subprocess
import sys
import threading
import time

HERE = os.path.dirname(os.path.abspath(__file__))
exe = os.path.join(HERE, "_build", os.environ.get("CCSM_UBENCH_EXE", "mfma_power_ceiling"))
secs = sys.argv[1] if len(sys.argv) > 1 else "4"
samples, stop = [], False


def num(d, key):
    for k, v in d.items():
        if key in k:
            try:
                return float(str(v).strip("()").replace("Mhz", ""))
            except ValueError:
                pass
    return float("nan")


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out)
            card = d[sorted(d)[0]]
            samples.append((time.time(), num(card, "ower"), num(card, "sclk clock speed")))
        except Exception:   # noqa: BLE001
            pass
        time.sleep(0.05)


threading.Thread(target=sampler, daemon=True).start()
p = subprocess.Popen([exe, secs], stdout=subprocess.PIPE, text=True)
begin = None
for line in p.stdout:
    now = time.time()
    if line.startswith("BEGIN"):
        begin = now
    elif line.startswith("END"):
        mine = [(w, c) for (t, w, c) in samples if begin + 0.5 * float(secs) < t < now and w == w and c == c]
        mine.sort()
        med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")   # noqa: E731
        print("%s | rocm-smi over the averaged half (%d samples): median %.0f W, sclk %.0f MHz" % (
            line.strip()[4:], len(mine), med([m[0] for m in mine]), med([m[1] for m in mine])), flush=True)
stop = True
sys.exit(p.wait())
