"""Aggregate model (config 5): NumPy oracle + torch.randn stream replica pinned against the reference's own outputs
(tests/golden/make_golden_aggr.py: real checkpoint, manual_seed(1234), reference functions)."""
import json
import os

import numpy as np

from conftest import GOLDEN
from ccsmeth_amd.utils import synth
from oracle import attbigru2s_oracle as orc
from oracle.torch_randn_replica import Mt19937Stream, normals_from_raw

G = np.load(os.path.join(GOLDEN, "aggr_golden.npz"))
META = json.load(open(os.path.join(GOLDEN, "aggr_golden.json")))
W = dict(np.load(os.path.join(GOLDEN, "aggr_ckpt_weights.npz")))
N_PARAMS = sum(v.size for v in W.values())      # = the 32-bit draws AggrAttRNN's construction consumes after the seed


def _stream(n_values):
    s = Mt19937Stream(META["seed"])
    s.raw(N_PARAMS)
    return normals_from_raw(s.raw(n_values))


def test_param_count_is_stream_offset():
    assert N_PARAMS == 14753


def test_randn_replica_matches_torch_draws():
    total = 64 * (1024 + 1024 + 376 + 700)
    n = _stream(total)
    assert np.abs(n[:65536].reshape(2, 1024, 32) - G["h0_first"]).max() < 1e-6
    assert np.abs(n[2 * 65536:2 * 65536 + 64 * 376].reshape(2, 376, 32) - G["h0_last_of_all"]).max() < 1e-6
    assert np.abs(n[2 * 65536 + 64 * 376:].reshape(2, 700, 32) - G["h0_hp1"]).max() < 1e-6
    assert (n[:65536].reshape(2, 1024, 32) == G["h0_first"]).mean() > 0.8      # ulp-level libm differences only


def test_ml_to_prob_and_histograms():
    assert [orc.cal_mod_prob(m) for m in range(256)] == G["ml2prob"].tolist()
    pile = synth.synth_pileup(META["n_pile"], META["pileup_seed"])
    pos, hist, cov = [], [], []
    for p, mls in zip(pile["pos"], pile["ml"]):
        probs = [orc.cal_mod_prob(int(m)) for m in mls]
        if len(probs) >= 4:
            pos.append(int(p)); hist.append(orc.normalized_histo(probs)); cov.append(len(probs))
    assert pos == G["pos"].tolist() and cov == G["covs"].tolist()
    assert np.array_equal(np.array(hist), G["histos"])


def test_aggregate_outputs_match_reference():
    n = _stream(64 * (2424 + 700))
    out, sp = orc.cal_modfreq_in_aggregate_mode(G["pos"], G["histos"], W, n, 0)
    assert sp == 64 * 2424
    assert np.abs(out - G["out_all"]).max() <= 2e-6
    a, b = META["sub"]
    out2, _ = orc.cal_modfreq_in_aggregate_mode(G["pos"][a:b], G["histos"][a:b], W, n, sp)
    assert np.abs(out2 - G["out_hp1"]).max() <= 2e-6
