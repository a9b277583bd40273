"""Read-level path: host NumPy extraction vs ccsm_forward_reads_host (GPU extraction + model) on synthetic HiFi reads.
env: NREADS (64), READLEN (15000), CPG (0.02 = fraction of positions made CG)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.utils import synth
from ccsmeth_amd import extract_features as ef

nr, L, cpg = int(os.environ.get("NREADS", "64")), int(os.environ.get("READLEN", "15000")), float(os.environ.get("CPG", "0.02"))
rng = np.random.default_rng(1)
reads = []
for i in range(nr):
    seq = rng.choice(list("ACGT"), size=L)
    for j in rng.integers(0, L - 1, int(L * cpg)):
        seq[j], seq[j + 1] = "C", "G"
    k = lambda: np.clip(rng.gamma(2.0, 20.0, size=L), 0, 255).astype(np.uint8)  # noqa: E731
    reads.append(("".join(seq), k(), k(), k(), k(), 12, 13))
t0 = time.perf_counter()
n = 0
for r in reads:
    n += len(ef.extract_read_arrays(*r[:5])["loc"])
t_host = time.perf_counter() - t0
dm = DeviceModel(synth.synth_weights(7), 0)
ws = dm.workspace(n)
ws.forward_reads(reads)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    first, locs, _, probs = ws.forward_reads(reads)
t_dev = (time.perf_counter() - t0) / reps
print("reads %d x %d bases, %d sites (%.1f per kb)" % (nr, L, n, 1000.0 * n / (nr * L)))
print("host NumPy extraction only : %.1f ms  (%.0f sites/s, 1 core)" % (1e3 * t_host, n / t_host))
print("ccsm_forward_reads_host    : %.1f ms  (%.0f sites/s incl. upload, extraction, model, download)" % (1e3 * t_dev, n / t_dev))
