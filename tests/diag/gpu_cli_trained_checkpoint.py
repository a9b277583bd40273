"""`call_mods` on a checkpoint TRAINED here: the [main]arithmetic log line of --arithmetic auto (what the probe measured and chose) and of a
forced split-mx-d.   usage: python tests/diag/gpu_cli_trained_checkpoint.py"""
import os, sys, subprocess, numpy as np, torch
from collections import OrderedDict
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from ccsmeth_amd.train import Trainer
from ccsmeth_amd.utils import synth, benchdata
n = 512
pool = synth.synth_sites(n * 8, 42)
lab = lambda s: (s["ipd1"][:, 10] + s["ipd2"][:, 10] > 0).astype(np.int64)
tr = Trainer(synth.synth_weights(41), device=0, max_sites=n)
for k in range(640):
    i = (k % 8) * n
    s = {key: v[i:i + n] for key, v in pool.items()}
    tr.forward_backward(s, lab(s), h0=None, dropout_rate=0.5, seed=41, step=k)
    tr.step(1e-3)
w = tr.state_dict(); tr.close()
torch.save(OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in w.items()), "/tmp/trained.ckpt")
print(benchdata.write_synthetic_hifi_bam("/tmp/cli_in.bam", 600, 15000))
for arith in ("auto", "split-mx-d"):
    r = subprocess.run([sys.executable, "-m", "ccsmeth_amd", "call_mods", "-i", "/tmp/cli_in.bam", "-m", "/tmp/trained.ckpt", "-o", "/tmp/cli_out_" + arith, "--arithmetic", arith, "--no_sort"],
                       cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True)
    print(arith, "rc", r.returncode)
    print("\n".join(l for l in (r.stdout + r.stderr).splitlines() if "arithmetic" in l or "sites/s" in l or "Error" in l)[:1500])
