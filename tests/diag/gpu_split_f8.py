"""split-f8 bring-up: selftest, parity vs NumPy oracle for a few sizes, per-kernel timing vs SPLIT3."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ccsmeth_amd import _lib
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.utils import synth
from oracle import attbigru2s_oracle as orc

lib = _lib.load()
a, b = C.c_float(), C.c_float()
_lib.check(lib.ccsm_selftest_split_f8(0, C.byref(a), C.byref(b)))
print("split-f8 selftest: err with corr %.3e, main only %.3e" % (a.value, b.value))
for wseed in (7, 11):
    w = synth.synth_weights(wseed)
    for n in (40, 333):
        s = synth.synth_sites(n, 8 + n); h1, h2 = synth.synth_h0(n, 9)
        ref = orc.attbigru2s_forward(w, s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h1, h2)[1]
        for prec in (3, 4):
            dm = DeviceModel(w, 0, precision=prec); ws = dm.workspace(n)
            _, p = ws.forward_host(s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h0=(h1, h2))
            print("wseed %d n %d precision %d: max |dprob| %.3e  finite %s" % (wseed, n, prec, np.abs(p - ref).max(), np.isfinite(p).all()))
            ws.close(); dm.close()
# timing, coalesced 3 x 2048
dev = torch.device("cuda:0")
w = synth.synth_weights(7)
s = synth.synth_sites(2048, 1); t = {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
args = (t["kmer1"], t["ipd1"], t["pw1"], t["npass1"], t["kmer2"], t["ipd2"], t["pw2"], t["npass2"])
for prec in (3, 4):
    dm = DeviceModel(w, 0, precision=prec); ws = dm.workspace(3 * 2048)
    outs = [(torch.empty((2048, 2), device=dev), torch.empty((2048, 2), device=dev)) for _ in range(3)]
    def step():
        for i in range(3):
            ws.group_add_torch(*args, out=outs[i], seed=1, offset=i * 2048)
        ws.group_run()
    for _ in range(5): step()
    torch.cuda.synchronize(); ws.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(40): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
    ms, nr = ws.timing_mean()
    print("precision %d: %.3f ms per 6144 sites (%.0f sites/s); kernels ms: %s (n=%d)" % (prec, dt * 1e3, 6144 / dt, np.round(ms, 4), nr))
