"""Layer 0 staggered against lock-step: the two forms must give the SAME BITS (same products in the same order), and the kernel times.
env: NSITES (default 6144 = 256 workgroups of 96 rows)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.utils import synth
n = int(os.environ.get("NSITES", "6144")); dev = torch.device("cuda:0")
os.environ["CCSM_WG_TILES"] = "3"
s = synth.synth_sites(n, 8); t = {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
args = (t["kmer1"], t["ipd1"], t["pw1"], t["npass1"], t["kmer2"], t["ipd2"], t["pw2"], t["npass2"])
bad = 0
for prec, name in ((4, "split-mx"), (3, "split3"), (5, "hybrid"), (6, "split-mx-d")):
    res = {}
    for form in ("lockstep", "stag"):
        if form == "lockstep": os.environ["CCSM_L0_LOCKSTEP"] = "1"
        else: os.environ.pop("CCSM_L0_LOCKSTEP", None)
        dm = DeviceModel(synth.synth_weights(7), 0, precision=prec)
        ws = dm.workspace(n)
        for _ in range(3): logits, probs = ws.forward_torch(*args, seed=5, offset=0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): ws.forward_torch(*args, seed=5, offset=0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        res[form] = (np.asarray(logits.cpu() if hasattr(logits, "cpu") else logits).copy(), dt)
    a, b = res["lockstep"][0], res["stag"][0]
    same = a.tobytes() == b.tobytes()
    bad += not same
    print("%-10s bits equal: %s   max |dlogit| %.3g   forward %.3f -> %.3f ms (%d sites)" % (name, same, float(np.abs(a - b).max()), res["lockstep"][1] * 1e3, res["stag"][1] * 1e3, n), flush=True)
sys.exit(1 if bad else 0)
