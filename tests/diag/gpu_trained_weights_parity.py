"""Train the model for a few hundred steps on a learnable synthetic labelling, then run the TRAINED parameters through the inference
library in both arithmetics against the oracle (h0 pinned and device-drawn-scale): is split-mx still within its bound on weights
that are not a synthetic initialisation?   usage: python tests/diag/gpu_trained_weights_parity.py [steps=320]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.train import Trainer
from ccsmeth_amd.utils import synth
from oracle import attbigru2s_oracle as orc

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 320
n = 512
pool = synth.synth_sites(n * 8, 42)
val = synth.synth_sites(n, 43)
lab = lambda s: (s["ipd1"][:, 10] + s["ipd2"][:, 10] > 0).astype(np.int64)  # noqa: E731


def train(wseed, nsteps):
    w0 = synth.synth_weights(wseed)
    tr = Trainer(w0, device=0, max_sites=n)
    for k in range(nsteps):
        i = (k % 8) * n
        s = {key: v[i:i + n] for key, v in pool.items()}
        loss, _ = tr.forward_backward(s, lab(s), h0=None, dropout_rate=0.5, seed=wseed, step=k)
        tr.step(1e-3)
    out = tr.state_dict()
    tr.close()
    return w0, out, loss


# what the probe of ccsm_create sees on independently trained checkpoints (and on their initialisations)
for wseed in (41, 5, 17, 23):
    for nsteps in (0, steps, 3 * steps):
        _, wt_, loss = train(wseed, nsteps) if nsteps else (None, synth.synth_weights(wseed), float("nan"))
        dm = DeviceModel(wt_, device=0)
        print("weights seed %2d, %4d steps (loss %.3f): probe selects %d   split-mx max %.2e tail %.4f   hybrid max %.2e tail %.4f" % (
            wseed, nsteps, loss, dm.precision, dm.probe_error, dm.probe_tail, dm.probe_error_hybrid, dm.probe_tail_hybrid))
        dm.close()
w, wt, loss = train(41, steps)
print("trained %d steps, last loss %.3f" % (steps, loss))
for k in ("rnn.weight_ih_l1", "rnn.weight_hh_l1", "rnn.bias_ih_l1", "_att3.Ua.weight", "fc1.weight"):
    print("  %-22s max |w| %.3f  rms %.4f   (initial: max %.3f rms %.4f)" % (k, np.abs(wt[k]).max(), np.sqrt((wt[k] ** 2).mean()), np.abs(w[k]).max(),
                                                                         np.sqrt((w[k] ** 2).mean())))
m = 256
sv = {k: v[:m] for k, v in val.items()}
for hseed, scale in ((99, 1.0), (98, 0.0)):
    h1, h2 = synth.synth_h0(m, hseed)
    h1, h2 = h1 * scale, h2 * scale
    _, ref = orc.attbigru2s_forward(wt, sv["kmer1"], sv["ipd1"], sv["pw1"], sv["npass1"], sv["kmer2"], sv["ipd2"], sv["pw2"], sv["npass2"], h1, h2)
    for prec in (0, 4, 5, 3):
        dm = DeviceModel(wt, device=0, precision=prec)
        ws = dm.workspace(m)
        _, probs = ws.forward_host(sv["kmer1"], sv["ipd1"], sv["pw1"], sv["npass1"], sv["kmer2"], sv["ipd2"], sv["pw2"], sv["npass2"], h0=(h1, h2))
        d = probs - ref
        print("h0 x%.0f precision %d -> in use %d  probe %.2e / hybrid %.2e quant %.3e : max |dprob| %.2e  mean |d| %.2e  mean d %+.2e   frac(prob>0.5) %.2f" % (
            scale, prec, dm.precision, dm.probe_error, dm.probe_error_hybrid, dm.quant_error, np.abs(d).max(), np.abs(d).mean(), d[:, 1].mean(), (ref[:, 1] > 0.5).mean()))
        ws.close(); dm.close()
if os.environ.get("SAVE_TRAINED"):
    np.savez_compressed(os.environ["SAVE_TRAINED"], **wt)
    print("saved", os.environ["SAVE_TRAINED"])
# a larger sample in the forced split-mx and hybrid arithmetics: how heavy is the tail?
big = synth.synth_sites(8192, 143)
h1, h2 = synth.synth_h0(8192, 144)
p = {}
for prec in (4, 5, 3):
    dm = DeviceModel(wt, device=0, precision=prec)
    ws = dm.workspace(8192)
    p[prec] = ws.forward_host(big["kmer1"], big["ipd1"], big["pw1"], big["npass1"], big["kmer2"], big["ipd2"], big["pw2"], big["npass2"], h0=(h1, h2))[1]
    ws.close(); dm.close()
for prec, name in ((4, "split-mx"), (5, "hybrid")):
    d = np.abs(p[prec] - p[3])[:, 1]
    print("8192 sites, %s vs split3: max %.2e  99.9%% %.2e  99%% %.2e  mean %.2e   sites above 1e-5: %d, above 5e-5: %d, above 1e-4: %d" % (
        name, d.max(), np.quantile(d, 0.999), np.quantile(d, 0.99), d.mean(), (d > 1e-5).sum(), (d > 5e-5).sum(), (d > 1e-4).sum()))
