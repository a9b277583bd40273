#!/usr/bin/env python
"""Golden fixtures for the training step (`ccsmeth trainm`), produced by the REFERENCE's model under
torch autograd on the CPU, exactly as train_multigpu.py:283-312 runs a step:

    outputs, _ = model(*16 tensors); loss = CrossEntropyLoss(weight=[1, pos_weight])(outputs, labels)
    optimizer.zero_grad(); loss.backward(); clip_grad_norm_(model.parameters(), 0.5); optimizer.step()   (Adam)

    python tests/golden/make_train_golden.py

Inputs (weights, sites, h0, labels) come from this repo's seeded generators (ccsmeth_amd/utils/synth.py); h0 is pinned by
wrapping torch.randn (the reference draws it per forward, models.py:77-87); dropout_rate = 0 (the only setting whose result
is defined without the reference's RNG stream).  Stored: loss, logits, per-tensor gradient norms, a seeded sample of 1024
gradient entries per tensor (all entries of small tensors), and the same sample of the parameters after 3 optimizer steps."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _ref_import import import_reference  # noqa: E402

import_reference()
import torch  # noqa: E402
import ccsmeth.models as ref_models  # noqa: E402

from ccsmeth_amd.utils import synth  # noqa: E402
from make_golden import PinnedRandn  # noqa: E402

torch.set_num_threads(8)
CASES = {"n48": dict(weight_seed=21, site_seed=301, h0_seed=401, label_seed=501, n=48, pos_weight=1.0),
         "n200_pw": dict(weight_seed=22, site_seed=302, h0_seed=402, label_seed=502, n=200, pos_weight=2.5)}
STEPS, LR, SAMPLE = 3, 1e-3, 1024


def step_inputs(case, k):
    """Batch k of a case: sites / h0 / labels from seeds offset by k."""
    n = case["n"]
    sites = synth.synth_sites(n, case["site_seed"] + 1000 * k)
    h1, h2 = synth.synth_h0(n, case["h0_seed"] + 1000 * k)
    labels = np.random.default_rng(case["label_seed"] + 1000 * k).integers(0, 2, n).astype(np.int64)
    return sites, h1, h2, labels


def forward(model, sites, h1, h2):
    n = sites["kmer1"].shape[0]
    f = lambda a: torch.tensor(np.asarray(a), dtype=torch.float)  # noqa: E731
    rep = lambda a: np.repeat(np.asarray(a)[:, None], 21, axis=1)  # noqa: E731
    z = f(np.zeros(n))
    with PinnedRandn([h1, h2]):
        return model(f(sites["kmer1"]), f(rep(sites["npass1"])), f(sites["ipd1"]), z, f(sites["pw1"]), z, z, z,
                     f(sites["kmer2"]), f(rep(sites["npass2"])), f(sites["ipd2"]), z, f(sites["pw2"]), z, z, z)


def main():
    out, meta = {}, {}
    for name, case in CASES.items():
        w = synth.synth_weights(case["weight_seed"])
        model = ref_models.ModelAttRNN(21, 3, 2, 0, 256, is_npass=True, is_sn=False, is_map=False, is_stds=False,
                                       model_type="attbigru2s", device=0)
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()})
        model.train()
        names = [k for k, _ in model.named_parameters()]
        assert names == list(w.keys()), (names, list(w.keys()))
        crit = torch.nn.CrossEntropyLoss(weight=torch.from_numpy(np.array([1, case["pos_weight"]])).float())
        opt = torch.optim.Adam(model.parameters(), lr=LR)
        losses, norms = [], []
        for k in range(STEPS):
            sites, h1, h2, labels = step_inputs(case, k)
            logits, _ = forward(model, sites, h1, h2)
            loss = crit(logits, torch.from_numpy(labels))
            opt.zero_grad()
            loss.backward()
            if k == 0:
                out[name + "_logits"] = logits.detach().numpy()
                for pn, p in model.named_parameters():
                    g = p.grad.detach().numpy().ravel()
                    out["%s_gnorm_%s" % (name, pn)] = np.array(np.linalg.norm(g.astype(np.float64)))
                    out["%s_g_%s" % (name, pn)] = g[synth.sample_index(pn, g.size, SAMPLE)]
            total = torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5)
            opt.step()
            losses.append(float(loss.detach()))
            norms.append(float(total))
        for pn, p in model.named_parameters():
            v = p.detach().numpy().ravel()
            out["%s_p_%s" % (name, pn)] = v[synth.sample_index(pn, v.size, SAMPLE)]
        meta[name] = dict(case, losses=losses, grad_norms=norms, steps=STEPS, lr=LR, sample=SAMPLE, param_names=names)
        print(name, losses, norms)
    # the initial parameters the reference draws after torch.manual_seed(tseed) (train_multigpu.py:470 + ModelAttRNN.__init__)
    torch.manual_seed(1234)
    m0 = ref_models.ModelAttRNN(21, 3, 2, 0.5, 256, is_npass=True, is_sn=False, is_map=False, is_stds=False,
                                model_type="attbigru2s", device=0)
    meta["init_tseed_1234"] = {k: dict(sum=float(v.detach().double().sum()), head=v.detach().numpy().ravel()[:4].astype(float).tolist())
                               for k, v in m0.named_parameters()}
    np.savez_compressed(os.path.join(HERE, "train_golden.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "train_golden.json"), "w"), indent=1, sort_keys=True)
    print(os.path.getsize(os.path.join(HERE, "train_golden.npz")))


if __name__ == "__main__":
    main()
