import os, sys
os.environ["CCSM_PHASE_DEBUG"] = "1"
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from ccsmeth_amd import _lib
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.utils import synth
n = 6144; dev = torch.device("cuda:0")
dm = DeviceModel(synth.synth_weights(7), 0, precision=4)
s = synth.synth_sites(n, 8); t = {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
ws = dm.workspace(n)
for _ in range(3): ws.forward_torch(t["kmer1"], t["ipd1"], t["pw1"], t["npass1"], t["kmer2"], t["ipd2"], t["pw2"], t["npass2"])
torch.cuda.synchronize()
buf = np.empty(21 * 8 * 5, np.uint64)
_lib.check(dm._lib.ccsm_debug_read(ws.handle, 5, buf.ctypes.data, buf.nbytes))
d = buf.reshape(21, 8, 5).astype(np.int64)[1:20]
print("C-end -> blend(tile0) done:", np.round((d[:, :, 1] - d[:, :, 3]).mean(0)))
print("blend(tile0) -> pack+stores(tile0) issued:", np.round((d[:, :, 2] - d[:, :, 1]).mean(0)))
print("tile0 done -> tail end (tiles 1,2):", np.round((d[:, :, 4] - d[:, :, 2]).mean(0)))
print("whole tail:", np.round((d[:, :, 4] - d[:, :, 3]).mean(0)))
