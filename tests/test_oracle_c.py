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


def test_c_oracle_instruction_set_clones_agree(corc, monkeypatch):
    """The AVX-512 / AVX2 / generic clones are the same arithmetic up to fused-multiply-add rounding."""
    w = synth.synth_weights(9)
    s = synth.synth_sites(50, 10)
    h1, h2 = synth.synth_h0(50, 11)
    args = (w, s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h1, h2)
    monkeypatch.delenv("ORACLE_ISA", raising=False)
    best = corc.forward(*args)[1]
    for isa in ("generic", "avx2"):
        monkeypatch.setenv("ORACLE_ISA", isa)
        if isa == "avx2" and " avx2 " not in open("/proc/cpuinfo").read():
            continue
        try:
            got = corc.forward(*args)[1]
        finally:
            monkeypatch.delenv("ORACLE_ISA", raising=False)
        assert np.abs(got - best).max() < 2e-6, isa
    assert corc.block_sites() % 12 == 0
