"""What a TRAINED checkpoint is served with, on committed weights (tests/golden/trained/*.npz; recipe tests/golden/make_trained_fixtures.py):
the gate VERDICT r03 asked for - deterministic (fixed weights, fixed inputs, deterministic kernels), and about the arithmetic `precision 0`
actually serves.  north_star: per-site probabilities within 1e-4 of the reference (fp32, models.py:125-130); this file holds the default to
a quarter of that on the two ordinary checkpoints at EVERY one of 16 x 8192 sites against the C oracle (BOUND below), and pins down
why the default is the three-pass arithmetic there: the block-scaled arithmetics' error on these weights is heavy-tailed
(profiles/r04_a_tail_study.log).  What "the reference" is worth on such weights: the C oracle (fp32, like the reference) itself sits
2.4e-6 / 2.5e-6 / 1.1e-5 from the float64 NumPy oracle on 1024 sites of the three checkpoints - two fp32 evaluations of a trained model
differ by that much - so a few 1e-6 against the C oracle is the noise floor of the comparison, not an error of this arithmetic.
The module sorts behind the other GPU modules on purpose (a failure here must not hide the training / end-to-end tests under `-x`)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from ccsmeth_amd.utils import synth

pytestmark = pytest.mark.gpu
FIXTURES = ["toy41_960", "planted7_5000", "planted11_12000_nodrop"]
# max |dprob| of the default arithmetic against the C oracle (fp32) over 16 x 8192 sites: a quarter of the bar (measured 7e-6).  The
# long-trained checkpoint (max |W_hh| 1.7) is held to the bar itself there, because fp32 evaluations of it differ by that much among
# themselves: over its first 1024 sites the C oracle sits 1.6e-5 from the float64 oracle, this library 9.7e-6, and over 131072 sites the
# two fp32 results are up to 6.8e-5 apart (measured, deterministic).  Against float64 arithmetic (1024 sites) every checkpoint is held
# to a quarter of the bar.
BOUND = {"toy41_960": 2.5e-5, "planted7_5000": 2.5e-5, "planted11_12000_nodrop": 1e-4}
BOUND64 = 2.5e-5
B = 8192


def _load(name):
    return dict(np.load(os.path.join(GOLDEN, "trained", name + ".npz")))


def _sites(b):
    """block b of the evaluation sites: plain synthetic sites and sites with the planted signal alternate"""
    return synth.synth_sites(B, 70000 + b) if b % 2 == 0 else synth.synth_labeled_sites(B, 70000 + b)[0]


def _args(s):
    return (s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"])


@pytest.mark.parametrize("name", FIXTURES)
def test_default_arithmetic_on_a_trained_checkpoint_holds_the_bar_at_every_site(name):
    """precision 0 on fixed trained weights: the same selection and probe figures on two creations; 16 x 8192 sites (explicit initial
    states) against the C oracle: none beyond BOUND (a quarter of the bar; the bar itself for the long-trained checkpoint, see BOUND); 1024
    of them against the float64 NumPy oracle as well (a quarter of the bar), next to the C oracle's own distance from it."""
    from ccsmeth_amd.models import DeviceModel
    from oracle import c_oracle
    from oracle import attbigru2s_oracle as orc
    wt = _load(name)
    dm = DeviceModel(wt, device=0)
    dm2 = DeviceModel(wt, device=0)
    assert (dm.precision, dm.probe_error, dm.probe_q999, dm.probe_sites) == (dm2.precision, dm2.probe_error, dm2.probe_q999, dm2.probe_sites)
    dm2.close()
    rule_ok = dm.probe_sites == 65536 and dm.probe_error <= 1.25e-5 and dm.probe_error <= 3.0 * dm.probe_q999
    assert dm.precision == (4 if rule_ok else 3), (dm.precision, dm.probe_error, dm.probe_q999, dm.probe_sites)
    assert dm.precision == 3 and dm.probe_error > 1.25e-5            # a trained checkpoint: split-mx's probe is not clean
    ws = dm.workspace(B)
    worst, n5, frac = 0.0, 0, []
    for b in range(16):
        s = _sites(b)
        h1, h2 = synth.synth_h0(B, 90000 + b)
        _, ref = c_oracle.forward(wt, *_args(s), h1, h2, threads=c_oracle.usable_threads())
        _, probs = ws.forward_host(*_args(s), h0=(h1, h2))
        d = np.abs(probs - ref)[:, 1]
        worst = max(worst, float(d.max()))
        n5 += int((d > 5e-5).sum())
        frac.append(float((ref[:, 1] > 0.5).mean()))
        if b == 1:                                                   # 1024 sites against float64 arithmetic
            q = {k: v[:1024] for k, v in s.items()}
            _, r64 = orc.attbigru2s_forward(wt, *_args(q), h1[:, :1024], h2[:, :1024])
            own64, c64 = float(np.abs(probs[:1024] - r64).max()), float(np.abs(ref[:1024] - r64).max())
    dm.close()
    print("%s: precision 0 -> %d, probe max %.2e (99.9 %% %.2e, %d sites); 16 x 8192 sites vs C oracle (fp32): max %.2e, beyond 5e-5: %d; "
          "1024 sites vs float64 oracle: %.2e (the C oracle itself: %.2e)" % (name, dm.precision, dm.probe_error, dm.probe_q999, dm.probe_sites, worst, n5, own64, c64))
    assert worst < BOUND[name] and own64 < BOUND64, (worst, n5, own64, c64)
    assert 0.05 < np.mean(frac) < 0.95                                # a model that discriminates


@pytest.mark.parametrize("name", FIXTURES[:2])
def test_block_scaled_arithmetics_are_heavy_tailed_on_trained_weights(name):
    """Why the probe does not hand trained checkpoints to split-mx / split-mx-d / the hybrid: over 16 x 8192 sites (device-drawn initial
    states, the same for every arithmetic) each of them, forced, leaves sites far beyond what its typical site shows - max / 99.9th
    percentile well above the 1.6 of the synthetic initialisation - and the fastest ones sites beyond 5e-5.  Deterministic: recorded in
    profiles/r04_a_tail_study.log for 2^20 sites; here the first 2^17."""
    from ccsmeth_amd.models import DeviceModel
    wt = _load(name)
    blocks = [_sites(b) for b in range(16)]

    def run(prec):
        dm = DeviceModel(wt, device=0, precision=prec)
        assert dm.precision == prec and dm.probe_sites == 0          # a forced precision is never overridden, no probe runs
        ws = dm.workspace(B)
        out = np.concatenate([ws.forward_host(*_args(s), h0=None, seed=777, offset=b * B)[1][:, 1] for b, s in enumerate(blocks)])
        dm.close()
        return out
    ref = run(3)
    res = {}
    for prec in (4, 6, 5):
        d = np.abs(run(prec) - ref)
        res[prec] = (float(d.max()), float(np.quantile(d, 0.999)), int((d > 1e-5).sum()), int((d > 5e-5).sum()))
    print(name, {k: "max %.2e q99.9 %.2e >1e-5 %d >5e-5 %d" % v for k, v in res.items()})
    assert res[4][0] > 5e-5 and res[4][3] > 0                       # split-mx: sites beyond half the bar
    assert res[5][2] <= res[6][2] <= res[4][2]                      # hybrid <= split-mx-d <= split-mx in sites beyond 1e-5
    for prec in (4, 6, 5):
        assert res[prec][0] > 1.25e-5                               # none of them would pass the probe's first condition


def test_synthetic_initialisation_keeps_split_mx_with_a_light_tail():
    """The other side of the rule: the benchmark's random initialisation (BASELINE config 2) passes the probe - 65536 sites, max within
    1.25e-5 and within 3 x the 99.9th percentile - and over 16 x 8192 further sites split-mx stays within 1.25e-5 of split3."""
    from ccsmeth_amd.models import DeviceModel
    for seed in (7, 20260928):
        w = synth.synth_weights(seed)
        dm = DeviceModel(w, device=0)
        assert dm.precision == 4 and dm.probe_sites == 65536 and dm.probe_error <= 1.25e-5 and dm.probe_error <= 3.0 * dm.probe_q999, (
            seed, dm.precision, dm.probe_error, dm.probe_q999, dm.probe_sites)
        ws = dm.workspace(B)
        d3 = DeviceModel(w, device=0, precision=3)
        w3 = d3.workspace(B)
        worst = 0.0
        for b in range(16 if seed == 7 else 2):
            s = _sites(b)
            a = ws.forward_host(*_args(s), h0=None, seed=777, offset=b * B)[1]
            r = w3.forward_host(*_args(s), h0=None, seed=777, offset=b * B)[1]
            worst = max(worst, float(np.abs(a - r).max()))
        dm.close(); d3.close()
        assert worst <= 1.25e-5, (seed, worst)
