"""Cycle stamps of the split-f8 attention kernel (workgroup 0, per wave): q projection, the three timestep groups (chunk loop /
tanh epilogue), softmax tail.  env CCSM_PHASE_DEBUG=1 CCSM_PHASE_LAYER=3 are set here."""
import os, sys
os.environ["CCSM_PHASE_DEBUG"] = "1"; os.environ["CCSM_PHASE_LAYER"] = "3"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ccsmeth_amd import _lib
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.utils import synth
n = 6144; dev = torch.device("cuda:0")
dm = DeviceModel(synth.synth_weights(7), 0, precision=4)
s = synth.synth_sites(n, 8); t = {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
ws = dm.workspace(n)
for _ in range(3):
    ws.forward_torch(t["kmer1"], t["ipd1"], t["pw1"], t["npass1"], t["kmer2"], t["ipd2"], t["pw2"], t["npass2"])
torch.cuda.synchronize()
buf = np.empty(8 * 16, np.uint64)
_lib.check(dm._lib.ccsm_debug_read(ws.handle, 5, buf.ctypes.data, buf.nbytes))
d = buf.reshape(8, 16).astype(np.int64)
names = ["q projection", "group 0 chunks", "group 0 tanh/e", "group 1 chunks", "group 1 tanh/e", "group 2 chunks", "group 2 tanh/e", "softmax tail"]
for i, nm in enumerate(names):
    print("%-16s" % nm, np.round(d[:, i + 1] - d[:, i]))
print("%-16s" % "total", np.round(d[:, 8] - d[:, 0]))
