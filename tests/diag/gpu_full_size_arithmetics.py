import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.utils import synth
n = 1_000_000
s = synth.synth_sites(n, 20260928)
w = synth.synth_weights(20260928)
out = {}
for prec in (3, 4, 5):
    dm = DeviceModel(w, device=0, precision=prec); ws = dm.workspace(8192)
    p = np.empty((n, 2), np.float32)
    for a in range(0, n, 8192):
        b = min(n, a + 8192); sub = {k: v[a:b] for k, v in s.items()}
        _, q = ws.forward_host(sub["kmer1"], sub["ipd1"], sub["pw1"], sub["npass1"], sub["kmer2"], sub["ipd2"], sub["pw2"], sub["npass2"], h0=None, seed=1234, offset=a)
        p[a:b] = q
    out[prec] = p; ws.close(); dm.close()
for prec in (4, 5):
    d = np.abs(out[prec][:, 1] - out[3][:, 1])
    print("precision %d vs split3 over %d sites: max %.2e  99.99%% %.2e  99.9%% %.2e  mean %.2e  >1e-5: %d" % (prec, n, d.max(), np.quantile(d, .9999), np.quantile(d, .999), d.mean(), (d > 1e-5).sum()))
