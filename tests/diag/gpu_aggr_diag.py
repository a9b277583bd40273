"""Aggregate kernel (ccsm_aggr.hip) against the NumPy oracle on small random regions: where do the errors sit?
usage: python tests/diag/gpu_aggr_diag.py [m=100]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ccsmeth_amd.call_mods_freq_bam import AggrModel
from oracle import attbigru2s_oracle as orc
from oracle.torch_randn_replica import Mt19937Stream, normals_from_raw

m = int(sys.argv[1]) if len(sys.argv) > 1 else 100
w = dict(np.load(os.path.join(ROOT, "tests", "golden", "aggr_ckpt_weights.npz")))
seed, n_params = 1234, 14753
s = Mt19937Stream(seed)
s.raw(n_params)
normals = normals_from_raw(s.raw(((2 * m * 64 + 15) // 16) * 16 + 64))
rng = np.random.default_rng(3)
pos = np.cumsum(rng.integers(2, 401, size=m)).astype(np.int64)
hist = np.zeros((m, 20), np.float32)
for i in range(m):
    h = np.histogram(rng.beta(0.3, 0.3, size=int(rng.integers(4, 61))), bins=20, range=[0, 1])[0]
    hist[i] = np.round(h / np.linalg.norm(h), 6)
model = AggrModel(w, device=0, tseed=seed, stream_sites=max(2 * m, 64))
for oc in ((False, True) if m <= 1024 else ()):
    got = model.forward_raw(pos, hist, only_close=oc) if oc else model.forward_raw(pos, hist)
    model.new_region()
    hm, pm = orc.aggregate_windows(pos, hist, 11, oc)
    h0 = np.asarray(normals[:64 * min(m, 1024)], np.float32).reshape(2, min(m, 1024), 32) if m <= 1024 else None
    want = orc.aggr_attbigru_forward(w, pm.astype(np.float32), hm.astype(np.float32), h0, dtype=np.float64)[:, 0]
    d = np.abs(got - want)
    print("only_close", oc, "m", m, "max err %.3e" % d.max(), "worst sites", np.argsort(-d)[:8], d[np.argsort(-d)[:8]])
    print("  got ", got[:6], "\n  want", want[:6])
# the reference's batching (1024 sites per h0 draw) and a second call on the same stream
model.new_region()
sp = 0
for (a, b) in ((0, m), (m // 3, m // 3 + m // 2)):
    got = model.forward_raw(pos[a:b], hist[a:b])
    want, sp2 = orc.cal_modfreq_in_aggregate_mode(pos[a:b], hist[a:b], w, normals, stream_pos=sp)
    d = np.abs(np.round(np.clip(got, 0, 1), 6) - want)
    print("call [%d, %d) stream_pos %d: max err %.3e, worst sites %s, %d beyond 1e-5" % (a, b, sp, d.max(), np.argsort(-d)[:6], (d > 1e-5).sum()))
    bad = np.flatnonzero(d > 1e-5)
    if len(bad):
        print("   first bad sites", bad[:20], "tiles", np.unique(bad // 32)[:20], "batches", np.unique(bad // 1024))
    sp = sp2
model.close()
# the reference's own outputs (tests/golden/aggr_golden.npz): fraction of bit-identical 6-dp values
import json
from ccsmeth_amd.call_mods_freq_bam import _cal_modfreq_in_aggregate_mode, _cal_mod_prob, _get_normalized_histo
from ccsmeth_amd.utils import synth
G = os.path.join(ROOT, "tests", "golden")
g = np.load(os.path.join(G, "aggr_golden.npz"))
meta = json.load(open(os.path.join(G, "aggr_golden.json")))
model = AggrModel({"module." + k: v for k, v in w.items()}, device=0, tseed=meta["seed"], stream_sites=1 << 14)
pile = synth.synth_pileup(meta["n_pile"], meta["pileup_seed"])
pos, hist = [], []
for p_, mls in zip(pile["pos"], pile["ml"]):
    probs = [_cal_mod_prob(int(x)) for x in mls]
    if len(probs) >= 4:
        pos.append(int(p_)); hist.append(_get_normalized_histo(probs))
model.new_region()
out_all = np.array(_cal_modfreq_in_aggregate_mode(pos, hist, model), np.float32)
print("golden: %d sites, max err %.2e, bit-identical 6-dp fraction %.4f" % (len(out_all), np.abs(out_all - g["out_all"]).max(), np.mean(out_all == g["out_all"])))
model.close()
