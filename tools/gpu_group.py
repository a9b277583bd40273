"""Run a few coalesced groups (3 x 2048 sites, one stream) — the launch shape bench.py times; for rocprofv3 --pmc passes."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.utils import synth
n = 2048; g = int(os.environ.get("COALESCE", "3")); dev = torch.device("cuda:0")
dm = DeviceModel(synth.synth_weights(7), 0, precision=int(os.environ.get("PREC", "3")))
s = synth.synth_sites(n * g, 8)
ws = dm.workspace(n * g)
batches = []
for b in range(g):
    sl = slice(b * n, (b + 1) * n)
    batches.append(tuple(torch.from_numpy(np.ascontiguousarray(s[k][sl])).to(dev) for k in
                         ("kmer1", "ipd1", "pw1", "npass1", "kmer2", "ipd2", "pw2", "npass2")))
for rep in range(int(os.environ.get("REPS", "4"))):
    for b in range(g):
        ws.group_add_torch(*batches[b], seed=1, offset=rep * n * g + b * n)
    ws.group_run()
torch.cuda.synchronize()
print("done")
