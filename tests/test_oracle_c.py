"""Pin the plain-C oracle (oracle/attbigru2s_oracle.c) against the reference's own outputs and the NumPy oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from ccsmeth_amd.utils import synth


@pytest.fixture(scope="module")
def corc():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    from oracle import c_oracle
    c_oracle.load()
    return c_oracle


@pytest.mark.parametrize("name", ["b21_n1", "b21_n64", "b21_n513", "b21_n2048"])
def test_c_oracle_matches_reference(corc, name):
    fwd = np.load(os.path.join(GOLDEN, "forward_golden.npz"))
    m = json.load(open(os.path.join(GOLDEN, "forward_golden.json")))[name]
    w = synth.synth_weights(m["weight_seed"])
    s = synth.synth_sites(m["n"], m["site_seed"])
    h1, h2 = synth.synth_h0(m["n"], m["h0_seed"])
    logits, probs = corc.forward(w, s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"],
                                 s["npass2"], h1, h2)
    assert np.abs(probs - fwd[name + "_probs"]).max() < 5e-6
    assert np.abs(logits - fwd[name + "_logits"]).max() < 2e-5


def test_c_oracle_thread_count_invariant(corc):
    w = synth.synth_weights(3)
    s = synth.synth_sites(40, 4)
    h1, h2 = synth.synth_h0(40, 5)
    a = corc.forward(w, s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h1, h2, threads=1)
    b = corc.forward(w, s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h1, h2, threads=4)
    assert np.array_equal(a[1], b[1])
