#!/usr/bin/env python
"""Golden fixtures for the aggregate model (BASELINE config 5), produced by the REFERENCE itself.

    python tests/golden/make_golden_aggr.py

Runs reference call_mods_freq_bam._get_normalized_histo / _cal_mod_prob / _cal_modfreq_in_aggregate_mode with the
reference's own AggrAttRNN and the only real checkpoint it ships
(models/model_ccsmeth_5mCpG_aggregate_attbigru_b11.v2p.ckpt), seeded exactly as
_call_modfreq_of_one_region_aggregate_mode does (torch.manual_seed(1234) BEFORE the model is constructed), on a synthetic
pile-up from this repo's generator.  Stores the checkpoint's tensors (61 KB: the only real weights available), the h0
draws torch produced, and the outputs."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _ref_import import import_reference, REF_ROOT  # noqa: E402

import_reference()
import torch  # noqa: E402
import ccsmeth.call_mods_freq_bam as fb  # noqa: E402
from ccsmeth.models import AggrAttRNN  # noqa: E402

from ccsmeth_amd.utils import synth  # noqa: E402

torch.set_num_threads(4)
CKPT = os.path.join(REF_ROOT, "models", "model_ccsmeth_5mCpG_aggregate_attbigru_b11.v2p.ckpt")


def main():
    para = torch.load(CKPT, map_location="cpu")
    weights = {k[7:] if k.startswith("module.") else k: v.numpy().astype(np.float32) for k, v in para.items()}
    np.savez_compressed(os.path.join(HERE, "aggr_ckpt_weights.npz"), **weights)
    print({k: v.shape for k, v in weights.items()}, "had module. prefix:", any(k.startswith("module.") for k in para))

    pile = synth.synth_pileup(2500, seed=555)          # positions, list of per-site ML bytes
    ml2prob = [fb._cal_mod_prob(m) for m in range(256)]
    # region caller order: seed, construct model (consumes RNG for parameter init), load checkpoint, eval
    torch.manual_seed(1234)
    model = AggrAttRNN(11, 1, 1, 0, 32, binsize=20, model_type="attbigru", device="cpu")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    model.eval()

    captured = []
    orig = torch.randn

    def cap(*shape, **kw):
        t = orig(*shape, **kw)
        captured.append(t.detach().numpy().copy())
        return t
    torch.randn = cap
    try:
        pos, histos, covs = [], [], []
        for p, mls in zip(pile["pos"], pile["ml"]):
            probs = [ml2prob[m] for m in mls]
            if len(probs) >= 4:
                pos.append(int(p))
                histos.append(fb._get_normalized_histo(probs, 4, 20))
                covs.append(len(probs))
        out_all = fb._cal_modfreq_in_aggregate_mode(pos, histos, model, 11, False)
        # a second call continues the same random stream (the reference calls all -> hp1 -> hp2)
        sub = slice(100, 100 + 700)
        out_hp1 = fb._cal_modfreq_in_aggregate_mode(pos[sub], histos[sub], model, 11, False)
    finally:
        torch.randn = orig
    print("sites", len(pos), "batches", len(captured), [c.shape for c in captured])
    store = dict(pos=np.array(pos, np.int64), histos=np.array(histos, np.float64), covs=np.array(covs, np.int64),
                 out_all=np.array(out_all, np.float32), out_hp1=np.array(out_hp1, np.float32),
                 h0_first=captured[0], h0_last_of_all=captured[2], h0_hp1=captured[3],
                 ml2prob=np.array(ml2prob, np.float64))
    np.savez_compressed(os.path.join(HERE, "aggr_golden.npz"), **store)
    json.dump(dict(pileup_seed=555, n_pile=2500, seed=1234, sub=[100, 800], n_sites=len(pos),
                   batches=[list(c.shape) for c in captured]), open(os.path.join(HERE, "aggr_golden.json"), "w"), indent=1)
    print(out_all[:8])


if __name__ == "__main__":
    main()
