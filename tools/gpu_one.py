"""Run a few single-stream forwards (for rocprofv3 counter collection).  env: PREC, NSITES, REPS"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.utils import synth
n = int(os.environ.get("NSITES", "2048")); dev = torch.device("cuda:0")
dm = DeviceModel(synth.synth_weights(7), 0, precision=int(os.environ.get("PREC", "3")))
s = synth.synth_sites(n, 8); t = {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
args = (t["kmer1"], t["ipd1"], t["pw1"], t["npass1"], t["kmer2"], t["ipd2"], t["pw2"], t["npass2"])
ws = dm.workspace(n)
for _ in range(int(os.environ.get("REPS", "4"))):
    ws.forward_torch(*args)
torch.cuda.synchronize()
print("done")
