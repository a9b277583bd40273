"""Training-step throughput of libccsm_train on synthetic batches (forward + backward + clip + Adam; dropout 0.5, device-drawn h0).
env: N (512), STEPS (60), WARMUP (10).  One JSON line: sites/s, ms/step and the share of upload / forward-backward / optimizer."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccsmeth_amd.train import Trainer
from ccsmeth_amd.utils import synth
n, steps, warm = int(os.environ.get("N", "512")), int(os.environ.get("STEPS", "60")), int(os.environ.get("WARMUP", "10"))
tr = Trainer(synth.synth_weights(41), device=0, max_sites=n)
s = synth.synth_sites(n, 42)
lab = (s["ipd1"][:, 10] + s["ipd2"][:, 10] > 0).astype(np.int64)
t_fb = t_st = 0.0
for k in range(warm + steps):
    if k == warm:
        t_fb = t_st = 0.0
        t0 = time.time()
    a = time.time()
    tr.forward_backward(s, lab, h0=None, dropout_rate=0.5, seed=1, step=k)
    b = time.time()
    tr.step(1e-3)
    c = time.time()
    t_fb += b - a; t_st += c - b
dt = time.time() - t0
flops = 3 * 244.23e6 * n * steps / dt        # backward ~ 2x forward
print(json.dumps(dict(metric="training sites/s (attbigru2s-b21, fp32)", value=n * steps / dt, batch=n, ms_per_step=dt / steps * 1e3,
                      ms_forward_backward=t_fb / steps * 1e3, ms_clip_adam=t_st / steps * 1e3, algorithmic_tflops=flops / 1e12)))
