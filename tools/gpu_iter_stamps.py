"""Per-iteration timeline of gru_layer12_mx16_kernel's phase A (workgroup 0, step 10, all 8 waves): cycle counter behind the barrier / behind the
previous pair's second half / behind the requests and the ring refill / behind the correction products / behind this pair's first half /
behind the counted wait / behind the next barrier.  Needs a -DCCSM_PHASE_STAMPS build named by CCSM_LIB_PATH; PREC=4."""
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
dm = DeviceModel(synth.synth_weights(7), 0, precision=4)
s = synth.synth_sites(n, 8); t = {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
args = (t["kmer1"], t["ipd1"], t["pw1"], t["npass1"], t["kmer2"], t["ipd2"], t["pw2"], t["npass2"])
ws = dm.workspace(n)
for _ in range(3):
    ws.forward_torch(*args)
torch.cuda.synchronize()
nb = 21 * 8 * 5 + 8 * 32 * 7
buf = np.empty(nb, np.uint64)
_lib.check(dm._lib.ccsm_debug_read(ws.handle, 5, buf.ctypes.data, buf.nbytes))
d = buf[21 * 8 * 5:].reshape(8, 32, 7).astype(np.int64)[:, 1:16, :]          # [wave][iteration 1..15][stamp]
seg = np.diff(d, axis=2)
names = ["2nd half i-1", "requests+refill", "corrections", "1st half i", "counted wait", "barrier"]
for par, label in ((0, "even iterations (behind an odd pair: 24 correction products)"), (1, "odd iterations")):
    print(label)
    print("   wave  " + "  ".join("%15s" % n_ for n_ in names) + "    iteration")
    for w in range(8):
        m = seg[w, (1 - par)::2].mean(0) if par == 0 else seg[w, 0::2].mean(0)
        it = (d[w, 2:, 0] - d[w, :-2, 0]).mean() / 2.0
        print("   w%d    " % w + "  ".join("%15.0f" % v for v in m) + "    %7.0f" % it)
if os.environ.get("VERBOSE"):
    print(seg[0]); print(seg[4])
