import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.utils import synth
n = 2048
dev = torch.device("cuda:0")
dm = DeviceModel(synth.synth_weights(7), 0, precision=3)
s = synth.synth_sites(n, 8)
t = {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
args = (t["kmer1"], t["ipd1"], t["pw1"], t["npass1"], t["kmer2"], t["ipd2"], t["pw2"], t["npass2"])
for timing in (False, True):
    for nstream in (1, 2, 4):
        wss = [dm.workspace(n) for _ in range(nstream)]
        for w in wss: w.set_timing(timing)
        streams = [torch.cuda.Stream(dev) for _ in range(nstream)]
        outs = [(torch.empty((n, 2), device=dev), torch.empty((n, 2), device=dev)) for _ in range(nstream)]
        steps = 96
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                k = i % nstream
                wss[k].forward_torch(*args, stream=streams[k].cuda_stream, out=outs[k], seed=1, offset=i * n)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print("timing=%s streams %d: %.3f ms/step (host issue %.3f ms/step) %.0f sites/s" % (timing, nstream, dt / steps * 1e3, (t1 - t0) / steps * 1e3, steps * n / dt))
        for x in wss: x.close()
dm.close()
