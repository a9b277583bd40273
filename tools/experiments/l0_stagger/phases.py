"""Slot timeline of the layer-0 kernel of workgroup 0 (cycle counter per wave), lock-step and staggered; needs the stamps build
(tools/ab_build.sh stamps -DCCSM_PHASE_STAMPS) named by CCSM_LIB_PATH.  Staggered stamps: 0 behind gamma (step start), 1 arrival at alpha,
2 arrival at beta, 3 behind beta, 4 arrival at gamma."""
import os, sys
os.environ["CCSM_PHASE_DEBUG"] = "1"; os.environ["CCSM_PHASE_LAYER"] = "0"; os.environ["CCSM_WG_TILES"] = "3"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
from ccsmeth_amd import _lib
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.utils import synth
n = 6144; dev = torch.device("cuda:0")
zero = os.environ.get("ZERO") == "1"
w = synth.synth_weights(7)
if zero: w = {k: np.zeros_like(v) for k, v in w.items()}
dm = DeviceModel(w, 0, precision=int(os.environ.get("PREC", "4")))
s = synth.synth_sites(n, 8); t = {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
args = (t["kmer1"], t["ipd1"], t["pw1"], t["npass1"], t["kmer2"], t["ipd2"], t["pw2"], t["npass2"])
ws = dm.workspace(n)
for _ in range(3): ws.forward_torch(*args)
torch.cuda.synchronize()
buf = np.empty(21 * 8 * 5 + 8, np.uint64)
_lib.check(dm._lib.ccsm_debug_read(ws.handle, 5, buf.ctypes.data, buf.nbytes))
d = buf[:21 * 8 * 5].reshape(21, 8, 5).astype(np.int64)
hw = buf[21 * 8 * 5:].astype(np.int64)
print("form:", "lock-step" if os.environ.get("CCSM_L0_LOCKSTEP") else "staggered", " zero weights" if zero else "")
print("HW_ID per wave: wave slot %s  SIMD %s  CU %s" % ([int(h & 15) for h in hw], [int((h >> 4) & 3) for h in hw], [int((h >> 8) & 15) for h in hw]))
ph = np.diff(d, axis=2)
print("cycles between stamps, mean over steps 2..18, per wave:")
for k in range(4): print("  %d->%d " % (k, k + 1), np.round(ph[2:19, :, k].mean(0)).astype(int))
print("  4->0' ", np.round((d[3:20, :, 0] - d[2:19, :, 4]).mean(0)).astype(int))
print("  step  ", np.round((d[3:20, :, 0] - d[2:19, :, 0]).mean(0)).astype(int))
t0 = d[5, 0, 0]
print("step 5 and 6, stamps relative to wave 0's step-5 start:")
for wv in range(8): print("  wave %d" % wv, (d[5, wv] - t0).tolist(), (d[6, wv] - t0).tolist())
