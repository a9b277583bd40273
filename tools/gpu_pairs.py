"""Per-pair timeline of gru_layer12_f3s_kernel's phases A and C (workgroup 0, step 10, all 8 waves): cycle counter at the pair's start /
before the wait for the x transfer / behind it / behind the barrier / at the pair's end.  Needs a -DCCSM_PHASE_STAMPS build named by
CCSM_LIB_PATH (tools/ab_build.sh stamps -DCCSM_PHASE_STAMPS); PREC=3."""
import os, sys
os.environ["CCSM_PHASE_DEBUG"] = "1"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ccsmeth_amd import _lib
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.utils import synth
n = int(os.environ.get("NSITES", "6144")); dev = torch.device("cuda:0")
dm = DeviceModel(synth.synth_weights(7), 0, precision=int(os.environ.get("PREC", "3")))
s = synth.synth_sites(n, 8); t = {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
args = (t["kmer1"], t["ipd1"], t["pw1"], t["npass1"], t["kmer2"], t["ipd2"], t["pw2"], t["npass2"])
ws = dm.workspace(n)
for _ in range(3):
    ws.forward_torch(*args)
torch.cuda.synchronize()
nb = 21 * 8 * 5 + 8 * 32 * 5
buf = np.empty(nb, np.uint64)
_lib.check(dm._lib.ccsm_debug_read(ws.handle, 5, buf.ctypes.data, buf.nbytes))
d = buf[21 * 8 * 5:].reshape(8, 32, 5).astype(np.int64)          # [wave][pair][stamp]
for name, sl in (("A", slice(0, 16)), ("C", slice(16, 32))):
    x = d[:, sl, :]
    seg = np.diff(x, axis=2)                                      # work before the wait | transfer wait | barrier | work behind it
    gap = x[:, 1:, 0] - x[:, :-1, 4]
    print("phase %s, cycles per pair (mean over its pairs 1..15; rows = waves 0..7):" % name)
    print("   before wait | vmcnt wait | barrier | behind barrier | to next pair | total")
    for w in range(8):
        m = seg[w, 1:].mean(0)
        print("   w%d  %7.0f %7.0f %7.0f %7.0f %7.0f  %7.0f" % (w, m[0], m[1], m[2], m[3], gap[w].mean(), (x[w, -1, 4] - x[w, 0, 0]) / 15.0))
    if os.environ.get("VERBOSE"):
        print(seg[0], seg[4])
