"""Host-side mirrors (feature extraction, batching, MM/ML) against fixtures produced by the reference itself."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from ccsmeth_amd import _bam2modbam as mm
from ccsmeth_amd import call_modifications as cm
from ccsmeth_amd import extract_features as ef
from ccsmeth_amd.utils import process_utils as pu

PIPE = np.load(os.path.join(GOLDEN, "pipeline_golden.npz"))
META = json.load(open(os.path.join(GOLDEN, "pipeline_golden.json")))


def test_tables():
    assert pu.codecv1_to_frame2() == PIPE["codecv1"].tolist()
    assert pu.base2code_dna == META["base2code_dna"]
    for s, rc in META["complement"].items():
        assert pu.complement_seq(s) == rc


@pytest.mark.parametrize("read", META["reads"], ids=lambda r: r["name"])
def test_extract_read_arrays_bit_exact(read):
    nm = read["name"]
    arr = ef.extract_read_arrays(read["seq"], PIPE[nm + "_fi"], PIPE[nm + "_ri"], PIPE[nm + "_fp"], PIPE[nm + "_rp"])
    assert arr is not None and len(arr["loc"]) == read["n_sites"]
    if read["n_sites"] == 0:
        return
    assert np.array_equal(arr["loc"], PIPE[nm + "_loc"])
    assert np.array_equal(arr["fkmer_ascii"], PIPE[nm + "_fkmer"]) and np.array_equal(arr["rkmer_ascii"], PIPE[nm + "_rkmer"])
    for k in ("fipd", "fpw", "ripd", "rpw"):
        assert arr[k].dtype == np.float64 and np.array_equal(arr[k], PIPE[nm + "_" + k]), k   # bit-exact float64
    rows = ef.extract_features_from_double_strand_read(nm, read["seq"], PIPE[nm + "_fi"], PIPE[nm + "_ri"], PIPE[nm + "_fp"],
                                                       PIPE[nm + "_rp"], read["fn"], read["rn"])
    assert len(rows) == read["n_sites"] and all(len(r) == 22 for r in rows)
    r0 = rows[0]
    assert r0[:5] == [".", -1, ".", nm, int(PIPE[nm + "_loc"][0])] and r0[6] == read["fn"] and r0[14] == read["rn"]
    assert r0[8] == "." and r0[12] == "." and r0[21] == 1 and r0[5][10:12] == "CG" and r0[13][10:12] == "CG"


def test_mismatched_kinetics_length_skips_read():
    assert ef.extract_read_arrays("ACGTACGTACGTACGTACGTACGTACGT", [1] * 5, [1] * 28, [1] * 28, [1] * 28) is None
    assert ef.extract_features_from_double_strand_read("x", "ACGT" * 7, [1] * 5, [1] * 28, [1] * 28, [1] * 28, 3, 3) == []


def _all_rows():
    rows = []
    for read in META["reads"]:
        nm = read["name"]
        rows += ef.extract_features_from_double_strand_read(nm, read["seq"], PIPE[nm + "_fi"], PIPE[nm + "_ri"],
                                                            PIPE[nm + "_fp"], PIPE[nm + "_rp"], read["fn"], read["rn"])
    return rows


def test_batch_feature_list2s_matches_reference():
    fb = cm._batch_feature_list2s(_all_rows())
    assert len(fb) == 18
    assert list(fb[0]) == META["sampleinfo"]
    assert np.array_equal(np.array(fb[1]), PIPE["batch_fkmers"]) and np.array_equal(np.array(fb[9]), PIPE["batch_rkmers"])
    assert np.array_equal(np.array(fb[2]), PIPE["batch_fpasss"]) and np.array_equal(np.array(fb[10]), PIPE["batch_rpasss"])
    assert fb[4][0] == 0 and fb[7][0] == 0 and fb[8][0] == 0     # "." placeholders become scalar 0
    assert np.array(fb[3]).dtype == np.float64


def test_mm_ml_and_refill():
    for case in META["mm_extra"]:
        if case["mm"] == "AssertionError":
            with pytest.raises(AssertionError):
                mm._convert_locs_to_mmtag(case["locs"], case["seq"])
        else:
            assert mm._convert_locs_to_mmtag(case["locs"], case["seq"]) == case["mm"]
    assert mm._convert_probs_to_mltag(META["ml_extra"]["probs"]) == META["ml_extra"]["ml"]
    reads = {r["name"]: r for r in META["reads"]}
    for name, exp in META["mmml"].items():
        lp = sorted((int(l), p) for l, p, h in zip(PIPE["pred_loc"], PIPE["pred_prob"], META["pred_holeid"]) if h == name)
        locs, probs = zip(*lp)
        assert mm._convert_locs_to_mmtag(locs, reads[name]["seq"]) == exp["mm"]
        assert mm._convert_probs_to_mltag(probs) == exp["ml"]
    tags = [("fi", [1, 2]), ("MM", "C+m,1;"), ("ML", [3]), ("np", 12), ("rp", [4]), ("sn", [1.0, 2.0])]
    j = lambda t: [list(x) for x in t]  # noqa: E731
    assert j(mm._refill_tags(tags, [3, 0], [10, 200], True)) == META["refill"]["rm"]
    assert j(mm._refill_tags(tags, [3, 0], [10, 200], False)) == META["refill"]["keep"]
    assert j(mm._refill_tags(tags, None, None, True)) == META["refill"]["none"]


def test_prob1_norm_round6_is_float32_rounding():
    p = np.array([[0.25, 0.75], [0.3333333, 0.6666667], [0.9999999, 1e-7]], np.float32)
    got = cm.prob1_norm_round6(p)
    exp = np.array([round(b / (a + b), 6) for a, b in p], np.float32)     # NumPy float32 scalar __round__, as the reference
    assert got.dtype == np.float32 and np.array_equal(got, exp)


def test_call_mods_cli_matches_reference_parser():
    """Every flag of the reference's `ccsmeth call_mods` (tests/golden/cli_golden.json, captured from the reference's own
    argparse) exists here with the same option strings and default; out-of-scope values are rejected with ValueError."""
    import argparse
    from ccsmeth_amd.call_mods import _check_scope, build_parser
    gold = json.load(open(os.path.join(GOLDEN, "cli_golden.json")))["flags"]
    ours = {a.dest: a for a in build_parser()._actions if a.dest != "help"}
    for dest, g in gold.items():
        assert dest in ours, dest
        assert sorted(ours[dest].option_strings) == sorted(g["options"]), dest
        assert ours[dest].default == g["default"], dest
        assert bool(ours[dest].required) == g["required"], dest
    base = ["-i", "a.bam", "-m", "m.ckpt", "-o", "out"]
    _check_scope(build_parser().parse_args(base + ["-p", "4", "--threads_call", "2", "--no_sort", "--keep_pulse"]))
    _check_scope(build_parser().parse_args(base + ["--mode", "align", "--mapq", "10", "--identity", "0.9", "--no_supplementary", "--skip_unmapped", "no"]))
    for norm in ("zscore", "min-mean", "min-max", "mad", "none"):        # every normalisation of the reference, raw codes too (host extraction)
        _check_scope(build_parser().parse_args(base + ["--norm", norm, "--no_decode"]))
    for extra in (["--seq_len", "20"], ["--mode", "reference"], ["--is_sn", "yes"], ["--model_type", "attbilstm2s"], ["--motifs", "CHG"],
                  ["--norm", "median"], ["--hid_rnn", "128"], ["--mode", "align", "--ref", "/nonexistent/g.fa"], ["--is_map", "yes"]):
        with pytest.raises(ValueError):
            _check_scope(build_parser().parse_args(base + extra))
    assert isinstance(build_parser(), argparse.ArgumentParser)


def test_every_kinetics_normalisation_matches_the_reference():
    """extract_features.py:181-199: none | zscore | min-max | min-mean on raw (--no_decode) and CodecV1-decoded codes against values
    generated by the reference's own function (tests/golden/make_norm_golden.py); a zero scale gives zeros.  `mad` against its
    definition (statsmodels.robust.scale.mad; statsmodels is not installed in the build container, so there is no run to pin it to)."""
    from ccsmeth_amd.extract_features import CODE2FRAMES, _normalize_signals
    g = np.load(os.path.join(GOLDEN, "norm_golden.npz"))
    for name in ("gamma", "flat", "short"):
        codes = g["codes_" + name]
        for dec in (0, 1):
            sig = CODE2FRAMES[codes] if dec else codes
            for method in ("none", "zscore", "min-max", "min-mean"):
                assert np.array_equal(np.asarray(_normalize_signals(sig, method), np.float64), g["%s_%s_dec%d" % (name, method, dec)]), (name, method, dec)
    x = CODE2FRAMES[g["codes_gamma"]].astype(np.float64)
    med = np.median(x)
    want = np.around((x - med) / (np.median(np.abs(x - med)) / 0.6744897501960817), 6)
    assert np.array_equal(_normalize_signals(x, "mad"), want)
    assert not np.any(_normalize_signals(CODE2FRAMES[g["codes_flat"]], "mad"))
    with pytest.raises(ValueError):
        _normalize_signals(x, "median")


def test_call_mods2s_coalesces_only_for_a_model_that_says_so():
    """_call_mods2s (call_modifications.py:170-227) against a stand-in model: a model with `coalesces_calls` gets launches of >=
    COALESCE_SITES sites and the reference's batch COUNT; any other model (the reference's own) gets the reference's cut; pinned initial
    states keep the cut either way; the per-site results are the same in all cases."""
    from ccsmeth_amd import call_modifications as cm

    class Fake:
        def __init__(self, coalesce):
            self.coalesces_calls = coalesce
            self.calls = []

        def __call__(self, kmer, *rest, h0=None):
            self.calls.append(len(kmer))
            p = (np.asarray(rest[1], np.float32)[:, 10] * 0.1 + 0.5).clip(0.01, 0.99)       # a function of the site's own features only
            probs = np.stack([1 - p, p], 1).astype(np.float32)
            return probs, probs
    n = 1300
    rng = np.random.default_rng(5)
    info = ["c\t%d\t+\th%d\t%d" % (i, i // 100, i) for i in range(n)]
    col = lambda: [rng.standard_normal(21) for _ in range(n)]  # noqa: E731
    fb = (info, *[col() for _ in range(16)], [0] * n)
    ref_model, co_model, pin_model = Fake(False), Fake(True), Fake(True)
    pred_ref, nb_ref = cm._call_mods2s(fb, ref_model, 512)
    pred_co, nb_co = cm._call_mods2s(fb, co_model, 512)
    pred_pin, nb_pin = cm._call_mods2s(fb, pin_model, 512, h0_provider=lambda b, k: None)
    assert ref_model.calls == [512, 512, 276] and co_model.calls == [1300] and pin_model.calls == [512, 512, 276]
    assert nb_ref == nb_co == nb_pin == 3
    assert pred_ref == pred_co == pred_pin
    assert cm._call_mods2s((info[:0], *[[] for _ in range(16)], []), Fake(True), 512) == ([], 0)
