"""Loss curve of libccsm_train on a learnable synthetic labelling (label = sign of the centre IPD sum). env: STEPS, RATE, LR, N."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccsmeth_amd.train import Trainer
from ccsmeth_amd.utils import synth
steps, rate, lr, n = int(os.environ.get("STEPS", "300")), float(os.environ.get("RATE", "0.5")), float(os.environ.get("LR", "1e-3")), int(os.environ.get("N", "512"))
w = synth.synth_weights(41)
tr = Trainer(w, device=0, max_sites=n)
pool = synth.synth_sites(n * 8, 42)
lab = lambda s: (s["ipd1"][:, 10] + s["ipd2"][:, 10] > 0).astype(np.int64)
val = synth.synth_sites(n, 43)
t0 = time.time()
for k in range(steps):
    i = (k % 8) * n
    s = {key: v[i:i + n] for key, v in pool.items()}
    loss, _ = tr.forward_backward(s, lab(s), h0=None, dropout_rate=rate, seed=1, step=k)
    gn = tr.step(lr)
    if k % 20 == 0 or k == steps - 1:
        ev, logits = tr.evaluate(val, lab(val), h0=None, seed=2, step=k)
        print("step %4d loss %.4f gnorm %.3f | val loss %.4f acc %.3f | %.1f ms/step" % (k, loss, gn, ev, (logits.argmax(1) == lab(val)).mean(), (time.time() - t0) * 1e3 / (k + 1)), flush=True)
