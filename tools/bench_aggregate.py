#!/usr/bin/env python
"""BASELINE config 5: aggregate attbigru_b11 over 50M synthetic pile-up sites on one MI355X (device-resident tables).
Processes `--regions` regions of `--region-sites` sites (each region restarts the seeded random stream, like the reference)
and prints one JSON line.  Secondary benchmark (bench.py is the contract line)."""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ccsmeth_amd import _lib
from ccsmeth_amd.call_mods_freq_bam import AggrModel

ap = argparse.ArgumentParser()
ap.add_argument("--regions", type=int, default=500)
ap.add_argument("--region-sites", type=int, default=100000)
a = ap.parse_args()
w = dict(np.load(os.path.join(ROOT, "tests", "golden", "aggr_ckpt_weights.npz")))
model = AggrModel(w, device=0, stream_sites=a.region_sites)
dev = torch.device("cuda:0")
rng = np.random.default_rng(5)
m = a.region_sites
pos = torch.from_numpy(np.cumsum(rng.integers(2, 401, size=m)).astype(np.int64)).to(dev)
cov = rng.integers(4, 61, size=m)
hist = np.zeros((m, 20), np.float32)
for i in range(m):
    h = np.histogram(rng.beta(0.3, 0.3, size=cov[i]), bins=20, range=[0, 1])[0]
    hist[i] = np.round(h / np.linalg.norm(h), 6)
hist_d = torch.from_numpy(hist).to(dev)
out = torch.empty(m, dtype=torch.float32, device=dev)
lib = model._lib
st = torch.cuda.current_stream(dev).cuda_stream
for _ in range(3):
    _lib.check(lib.ccsm_aggr_forward_device(model.handle, m, pos.data_ptr(), hist_d.data_ptr(), 0, out.data_ptr(), st))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.regions):
    _lib.check(lib.ccsm_aggr_forward_device(model.handle, m, pos.data_ptr(), hist_d.data_ptr(), 0, out.data_ptr(), st))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
sites = a.regions * m
print(json.dumps({"metric": "aggregate-mode sites/s (attbigru_b11, config 5)", "value": sites / dt, "unit": "sites/s",
                  "sites": sites, "seconds": dt, "flops_per_site": 275.3e3, "TFLOPs": sites / dt * 275.3e3 / 1e12,
                  "algorithmic_GBps": sites / dt * 88 / 1e9, "reference_cpu_sites_per_s_8thr": 73500}))
