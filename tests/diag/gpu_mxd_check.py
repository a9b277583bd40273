"""split-mx-d (precision 6: fp6 recurrent weight blobs + per-(row, 32-k) dynamic scales of the state's correction blob) on TRAINED
checkpoints: max |dprob| and tail counts over 8192 sites against split3, beside split-mx (4) and the hybrid (5), for several
independently trained checkpoints; and what ccsm_create's probe makes of each.   usage: python tests/diag/gpu_mxd_check.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.train import Trainer
from ccsmeth_amd.utils import synth

n = 512
pool = synth.synth_sites(n * 8, 42)
lab = lambda s: (s["ipd1"][:, 10] + s["ipd2"][:, 10] > 0).astype(np.int64)  # noqa: E731


def train(wseed, nsteps):
    tr = Trainer(synth.synth_weights(wseed), device=0, max_sites=n)
    for k in range(nsteps):
        i = (k % 8) * n
        s = {key: v[i:i + n] for key, v in pool.items()}
        tr.forward_backward(s, lab(s), h0=None, dropout_rate=0.5, seed=wseed, step=k)
        tr.step(1e-3)
    out = tr.state_dict()
    tr.close()
    return out


big = synth.synth_sites(8192, 143)
h1, h2 = synth.synth_h0(8192, 144)
cases = [("synthetic 7", synth.synth_weights(7)), ("heavy 7", synth.synth_weights_heavy(7)), ("heavy 11", synth.synth_weights_heavy(11))]
cases += [("trained %d/%d" % (sd, st), train(sd, st)) for sd, st in ((41, 960), (5, 320), (17, 640), (23, 960))]
for name, wt in cases:
    p = {}
    for prec in (3, 4, 6, 5):
        dm = DeviceModel(wt, device=0, precision=prec)
        ws = dm.workspace(8192)
        p[prec] = ws.forward_host(big["kmer1"], big["ipd1"], big["pw1"], big["npass1"], big["kmer2"], big["ipd2"], big["pw2"], big["npass2"], h0=(h1, h2))[1]
        ws.close(); dm.close()
    dm = DeviceModel(wt, device=0)
    print("%-16s probe selects %d (split-mx %.1e / %.4f, split-mx-d %.1e / %.4f, hybrid %.1e / %.4f)" % (
        name, dm.precision, dm.probe_error, dm.probe_tail, dm.probe_error_mxd, dm.probe_tail_mxd, dm.probe_error_hybrid, dm.probe_tail_hybrid))
    dm.close()
    for prec, nm in ((4, "split-mx"), (6, "split-mx-d"), (5, "hybrid")):
        d = np.abs(p[prec] - p[3])[:, 1]
        print("    %-10s vs split3 over 8192 sites: max %.2e  99.9%% %.2e  mean %.2e   beyond 1e-5: %d, 5e-5: %d, 1e-4: %d" % (
            nm, d.max(), np.quantile(d, 0.999), d.mean(), (d > 1e-5).sum(), (d > 5e-5).sum(), (d > 1e-4).sum()))
