#!/usr/bin/env python
"""Quick per-kernel timing + stream-concurrency probe (GPU box).  python tools/gpu_timing.py [precision] [n_sites]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from ccsmeth_amd.models import DeviceModel  # noqa: E402
from ccsmeth_amd.utils import synth  # noqa: E402

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
dev = torch.device("cuda:0")
w = synth.synth_weights(7)
dm = DeviceModel(w, 0, precision=prec)
s = synth.synth_sites(n, 8)
t = {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
args = (t["kmer1"], t["ipd1"], t["pw1"], t["npass1"], t["kmer2"], t["ipd2"], t["pw2"], t["npass2"])

ws = dm.workspace(n)
ws.set_timing(True)
for _ in range(3):
    ws.forward_torch(*args)
torch.cuda.synchronize()
tm = np.mean([(ws.forward_torch(*args), torch.cuda.synchronize(), ws.last_timing())[2] for _ in range(5)], axis=0)
print("precision", prec, "n", n, "kernel ms [gru0, gru1, gru2, attn, misc]:", np.round(tm, 4), "sum", round(float(tm.sum()), 4))
fl = n * np.array([34.45e6, 99.09e6, 99.09e6, 11.6e6])
print("TFLOP/s (algorithmic) per kernel:", np.round(fl / (tm[:4] * 1e-3) / 1e12, 1), " single-stream sites/s: %.0f" % (n / (tm.sum() * 1e-3)))
ws.set_timing(False)

for nstream in (1, 2, 3, 4, 6, 8):
    wss = [dm.workspace(n) for _ in range(nstream)]
    streams = [torch.cuda.Stream(dev) for _ in range(nstream)]
    outs = [(torch.empty((n, 2), device=dev), torch.empty((n, 2), device=dev)) for _ in range(nstream)]
    steps = 48
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            k = i % nstream
            wss[k].forward_torch(*args, stream=streams[k].cuda_stream, out=outs[k], seed=1, offset=i * n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("streams %d: %.3f ms/step  %.0f sites/s" % (nstream, dt / steps * 1e3, steps * n / dt))
    for x in wss:
        x.close()
dm.close()
