"""Phase timeline of one split-mx GRU layer (workgroup 0, all 8 waves): cycle counter at step start / after phase A / B / C / tail.
Needs a library built with -DCCSM_PHASE_STAMPS (tools/ab_build.sh stamps -DCCSM_PHASE_STAMPS) named by CCSM_LIB_PATH."""
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
buf = np.empty(21 * 8 * 5, np.uint64)
_lib.check(dm._lib.ccsm_debug_read(ws.handle, 5, buf.ctypes.data, buf.nbytes))
d = buf[:21 * 8 * 5].reshape(21, 8, 5).astype(np.int64)
ph = np.diff(d, axis=2)                       # [step][wave][A, B, C, epilogue]
gap = d[1:, :, 0] - d[:-1, :, 4]
names = os.environ.get("PHASE_NAMES", "X,H,sigm,tail").split(",")      # one-pass kernel; two-pass build: A,B,C,tail
print("cycles per phase (mean over steps 1..19, per wave):")
for k, nm in enumerate(names):
    print("  %-5s" % nm, np.round(ph[1:20, :, k].mean(0)))
print("  gap  ", np.round(gap[1:19].mean(0)))
print("  step ", np.round((d[2:20, :, 0] - d[1:19, :, 0]).mean(0)))
print("ideal MFMA cycles per wave and step (x2 waves per SIMD): 64-row one-pass kernel X 16 x (12 x 32 + 6 x 32) = 9216, H 8 x 576 = 4608")
