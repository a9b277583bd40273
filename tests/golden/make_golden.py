#!/usr/bin/env python
"""Generate the committed golden fixtures by running the REFERENCE itself (build container only).

    python tests/golden/make_golden.py

Imports /root/reference/ccsmeth read-only (tests/golden/_ref_import.py), feeds it inputs drawn from this
repo's own seeded generators (ccsmeth_amd/utils/synth.py), pins h0 by wrapping torch.randn, and stores
inputs' seeds + the reference's outputs as small .npz/.json files next to this script.  Fixtures are data:
no reference source text is stored.
"""
import json
import os
import sys
from argparse import Namespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from _ref_import import import_reference  # noqa: E402

import_reference()
import torch  # noqa: E402
import ccsmeth.models as ref_models  # noqa: E402
import ccsmeth.call_modifications as ref_cm  # noqa: E402
import ccsmeth.extract_features as ref_ef  # noqa: E402
import ccsmeth._bam2modbam as ref_mm  # noqa: E402
import ccsmeth.utils.process_utils as ref_pu  # noqa: E402

from ccsmeth_amd.utils import synth  # noqa: E402

torch.set_num_threads(8)


class PinnedRandn:
    """Replace torch.randn during a forward so init_hidden (models.py:77-87) returns our h0, strand 1 first."""

    def __init__(self, tensors):
        self.q = list(tensors)
        self.orig = torch.randn

    def __enter__(self):
        def fake(*shape, **kw):
            t = self.q.pop(0)
            assert tuple(t.shape) == tuple(shape), (t.shape, shape)
            return torch.from_numpy(np.ascontiguousarray(t)).clone().requires_grad_(kw.get("requires_grad", False))
        torch.randn = fake
        return self

    def __exit__(self, *a):
        torch.randn = self.orig


def build_ref_model(weights, num_layers, hidden):
    model = ref_models.ModelAttRNN(21, num_layers, 2, 0, hidden, is_npass=True, is_sn=False, is_map=False,
                                   is_stds=False, model_type="attbigru2s", device=0)
    sd = {k: torch.from_numpy(v.copy()) for k, v in weights.items()}
    model.load_state_dict(sd)
    model.eval()
    return model


def ref_forward(model, sites, h0_1, h0_2):
    n = sites["kmer1"].shape[0]
    f = lambda a: torch.tensor(np.asarray(a), dtype=torch.float)  # noqa: E731  (constants_torch.FloatTensor on CPU)
    rep = lambda a: np.repeat(np.asarray(a)[:, None], 21, axis=1)  # noqa: E731
    zeros = f(np.zeros(n))
    with PinnedRandn([h0_1, h0_2]):
        logits, probs = model(f(sites["kmer1"]), f(rep(sites["npass1"])), f(sites["ipd1"]), zeros, f(sites["pw1"]), zeros,
                              zeros, zeros,
                              f(sites["kmer2"]), f(rep(sites["npass2"])), f(sites["ipd2"]), zeros, f(sites["pw2"]), zeros,
                              zeros, zeros)
    return logits.detach().numpy(), probs.detach().numpy()


def gen_forward():
    cases = [  # name, weight seed, site seed, h0 seed, n, layers, hidden
        ("b21_n1", 11, 101, 201, 1, 3, 256),
        ("b21_n64", 11, 102, 202, 64, 3, 256),
        ("b21_n513", 12, 103, 203, 513, 3, 256),
        ("small_n37", 13, 104, 204, 37, 2, 32),
        ("b21_n2048", 14, 105, 205, 2048, 3, 256),      # BASELINE.json configs[1] batch size, full model
    ]
    out = {}
    meta = {}
    for name, ws, ss, hs, n, layers, hidden in cases:
        w = synth.synth_weights(ws, num_layers=layers, hidden=hidden)
        sites = synth.synth_sites(n, ss)
        h1, h2 = synth.synth_h0(n, hs, num_layers=layers, hidden=hidden)
        model = build_ref_model(w, layers, hidden)
        logits, probs = ref_forward(model, sites, h1, h2)
        out[name + "_logits"] = logits
        out[name + "_probs"] = probs
        meta[name] = dict(weight_seed=ws, site_seed=ss, h0_seed=hs, n=n, num_layers=layers, hidden=hidden)
        print(name, probs[:2])
    np.savez_compressed(os.path.join(HERE, "forward_golden.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "forward_golden.json"), "w"), indent=1, sort_keys=True)


VARIANTS = [  # name, (is_npass, is_stds, is_sn, is_map): combinations that differ from the default, the 17- and 18-column ones included
    ("sn", (True, False, True, False)), ("map", (True, False, False, True)), ("stds", (True, True, False, False)),
    ("stds_map", (True, True, False, True)), ("sn_map", (True, False, True, True)), ("nonpass", (False, False, False, False)),
    ("nonpass_stds_sn", (False, True, True, False)), ("nonpass_stds_sn_map", (False, True, True, True)),
    ("stds_sn", (True, True, True, False)), ("stds_sn_map", (True, True, True, True)),       # 17 and 18 input columns
]


def gen_forward_variants():
    """ModelAttRNN built with is_stds / is_sn / is_map / without is_npass (models.py:39-47, 100-123): the optional feature planes."""
    out, meta = {}, {}
    n = 48
    for k, (name, (is_npass, is_stds, is_sn, is_map)) in enumerate(VARIANTS):
        ws, ss, hs, es = 31 + k, 301 + k, 401 + k, 501 + k
        w = synth.synth_weights(ws, feas_ccs=synth.feas_ccs_of(is_npass, is_stds, is_sn, is_map))
        sites = synth.synth_sites(n, ss)
        h1, h2 = synth.synth_h0(n, hs)
        ex = synth.synth_extras(n, es, is_stds, is_sn, is_map)
        model = ref_models.ModelAttRNN(21, 3, 2, 0, 256, is_npass=is_npass, is_sn=is_sn, is_map=is_map, is_stds=is_stds,
                                       model_type="attbigru2s", device=0)
        model.load_state_dict({k2: torch.from_numpy(v.copy()) for k2, v in w.items()})
        model.eval()
        f = lambda a: torch.tensor(np.asarray(a), dtype=torch.float)  # noqa: E731
        rep = lambda a: np.repeat(np.asarray(a)[:, None], 21, axis=1)  # noqa: E731
        zeros = f(np.zeros(n))
        args = []
        for s_, e in (("1", ex[0]), ("2", ex[1])):
            args += [f(sites["kmer" + s_]), f(rep(sites["npass" + s_])), f(sites["ipd" + s_]), f(e["ipd_std"]) if is_stds else zeros,
                     f(sites["pw" + s_]), f(e["pw_std"]) if is_stds else zeros, f(e["sn"]) if is_sn else zeros, f(e["map"]) if is_map else zeros]
        with PinnedRandn([h1, h2]):
            logits, probs = model(*args)
        out[name + "_logits"], out[name + "_probs"] = logits.detach().numpy(), probs.detach().numpy()
        meta[name] = dict(weight_seed=ws, site_seed=ss, h0_seed=hs, extra_seed=es, n=n, is_npass=is_npass, is_stds=is_stds, is_sn=is_sn,
                          is_map=is_map)
        print(name, probs[:2])
    np.savez_compressed(os.path.join(HERE, "forward_variants_golden.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "forward_variants_golden.json"), "w"), indent=1, sort_keys=True)


class FakeRead:
    """Duck-typed pysam.AlignedSegment exposing what extract_features.py:88-126 reads."""

    def __init__(self, name, seq, fi, ri, fp, rp, fn, rn, is_reverse=False):
        self.query_name = name
        self._seq = seq
        self.query_alignment_start = 0
        self.query_alignment_end = len(seq)
        self.reference_name = None
        self.reference_start = -1
        self.reference_end = None
        self.cigartuples = None
        self.flag = 4
        self.mapping_quality = 255
        self.is_unmapped = True
        self.is_secondary = False
        self.is_duplicate = False
        self.is_supplementary = False
        self.is_reverse = is_reverse
        self._tags = dict(fi=fi, ri=ri, fp=fp, rp=rp, fn=fn, rn=rn, sn=[10.0, 11.0, 12.0, 13.0])

    def get_forward_sequence(self):
        return self._seq

    def get_cigar_stats(self):
        return [[0] * 11, [0] * 11]

    def get_tag(self, k):
        return self._tags[k]


def synth_read(rng, length, cg_every=None):
    seq = rng.choice(list("ACGT"), size=length)
    if cg_every:
        for i in range(5, length - 1, cg_every):
            seq[i], seq[i + 1] = "C", "G"
    seq = "".join(seq)
    codes = lambda: np.clip(rng.gamma(2.0, 14.0, size=length), 0, 255).astype(np.uint8)  # noqa: E731
    return seq, codes(), codes(), codes(), codes(), int(rng.integers(3, 31)), int(rng.integers(3, 31))


def gen_extract_and_pipeline():
    args = Namespace(mode="denovo", no_decode=False, norm="zscore", is_sn="no", is_map="no", seq_len=21,
                     mod_loc=0, methy_label=1, skip_unmapped="yes", no_supplementary=False, mapq=0, identity=0.0)
    rng = np.random.default_rng(4242)
    reads = []
    specs = [("r0", 300, 17), ("r1", 1000, None), ("r2", 25, 7), ("r3", 20, 3), ("r4", 640, 40)]
    for name, length, cg in specs:
        reads.append((name,) + synth_read(rng, length, cg))
    # r5: constant kinetics -> std == 0 branch of _normalize_signals; ends in CG (edge sites rejected)
    seq5 = "ACGTTACGGACGTTTACGCGATATCGCGAATTCGACGTACGATCGATTACGCG"
    n5 = len(seq5)
    reads.append(("r5", seq5, np.full(n5, 7, np.uint8), np.full(n5, 200, np.uint8), np.full(n5, 0, np.uint8),
                  np.full(n5, 255, np.uint8), 5, 9))
    store = {}
    meta = {"reads": []}
    feature_list = []
    holeidxes = []
    for ridx, (name, seq, fi, ri, fp, rp, fn, rn) in enumerate(reads):
        fr = FakeRead(name, seq, fi, ri, fp, rp, fn, rn)
        rows = ref_ef.extract_features_from_double_strand_read(fr, ["CG"], None, None, None, args)
        meta["reads"].append(dict(name=name, seq=seq, fn=fn, rn=rn, n_sites=len(rows)))
        for t, arr in (("fi", fi), ("ri", ri), ("fp", fp), ("rp", rp)):
            store[f"{name}_{t}"] = np.asarray(arr, np.uint8)
        if rows:
            store[f"{name}_loc"] = np.array([r[4] for r in rows], np.int64)
            store[f"{name}_fkmer"] = np.array([[ord(c) for c in r[5]] for r in rows], np.uint8)
            store[f"{name}_fipd"] = np.array([r[7] for r in rows], np.float64)
            store[f"{name}_fpw"] = np.array([r[9] for r in rows], np.float64)
            store[f"{name}_rkmer"] = np.array([[ord(c) for c in r[13]] for r in rows], np.uint8)
            store[f"{name}_ripd"] = np.array([r[15] for r in rows], np.float64)
            store[f"{name}_rpw"] = np.array([r[17] for r in rows], np.float64)
            for r in rows:
                assert r[0] == "." and r[1] == -1 and r[2] == "." and r[3] == name and r[6] == fn and r[14] == rn
                assert r[8] == "." and r[10] == "." and r[11] == "." and r[12] == "."
        feature_list += rows
        holeidxes += [ridx] * len(rows)
        print(name, len(seq), "sites:", len(rows))

    # _batch_feature_list2s (call_modifications.py:73-123) + _call_mods2s (:170-227) with pinned h0, batch 64
    fb = ref_cm._batch_feature_list2s(feature_list)
    n = len(feature_list)
    store["batch_fkmers"] = np.array(fb[1], np.int64)
    store["batch_fpasss"] = np.array(fb[2], np.int64)
    store["batch_rkmers"] = np.array(fb[9], np.int64)
    store["batch_rpasss"] = np.array(fb[10], np.int64)
    meta["sampleinfo"] = list(fb[0])
    meta["holeidxes"] = holeidxes
    w = synth.synth_weights(21)
    model = build_ref_model(w, 3, 256)
    bs = 64
    h0s = []
    hseed = 900
    for i in range(0, n, bs):
        m = min(bs, n - i)
        h1, h2 = synth.synth_h0(m, hseed + i // bs)
        h0s += [h1, h2]
    ref_cm.use_cuda = False
    with PinnedRandn(h0s):
        pred_info, batch_num = ref_cm._call_mods2s(fb, model, bs, 0)
    meta["call_mods"] = dict(weight_seed=21, batch_size=bs, h0_seed_base=hseed, batch_num=int(batch_num))
    store["pred_prob"] = np.array([p[2] for p in pred_info], np.float32)
    store["pred_loc"] = np.array([p[1] for p in pred_info], np.int64)
    meta["pred_holeid"] = [p[0] for p in pred_info]

    # MM / ML per read (call_modifications.py:230-263 logic, _bam2modbam.py:187-226)
    mm = {}
    for ridx, (name, seq, *_rest) in enumerate(reads):
        lp = sorted([(p[1], p[2]) for p in pred_info if p[0] == name])
        if not lp:
            continue
        locs, probs = zip(*lp)
        mm[name] = dict(mm=ref_mm._convert_locs_to_mmtag(locs, seq), ml=ref_mm._convert_probs_to_mltag(probs))
        # reverse-strand read: seq_fwd = revcomp(query_sequence)  (call_modifications.py:245)
    meta["mmml"] = mm
    # extra MM/ML cases incl. assertion failure and p>=1
    extra = []
    for locs, seq in (([1, 7], "ACGTTTTCG"), ([1], "ACGTTTTCG"), ([7], "ACGTTTTCG"), ([0, 2, 4], "CCCCCC")):
        extra.append(dict(locs=locs, seq=seq, mm=ref_mm._convert_locs_to_mmtag(locs, seq)))
    for locs, seq in (([2], "ACGT"), ([], "ACGT")):
        try:
            ref_mm._convert_locs_to_mmtag(locs, seq)
            extra.append(dict(locs=locs, seq=seq, mm="no-error"))
        except AssertionError:
            extra.append(dict(locs=locs, seq=seq, mm="AssertionError"))
    meta["mm_extra"] = extra
    probs = [0.0, 0.0039062, 0.00390625, 0.5, 0.999999, 1.0, 1.5, np.float32(0.996094), np.float32(0.3)]
    meta["ml_extra"] = dict(probs=[float(p) for p in probs], ml=ref_mm._convert_probs_to_mltag(probs))
    tags = [("fi", [1, 2]), ("MM", "C+m,1;"), ("ML", [3]), ("np", 12), ("rp", [4]), ("sn", [1.0, 2.0])]
    meta["refill"] = dict(
        rm=[list(map(_j, t)) for t in ref_mm._refill_tags(tags, [3, 0], [10, 200], True)],
        keep=[list(map(_j, t)) for t in ref_mm._refill_tags(tags, [3, 0], [10, 200], False)],
        none=[list(map(_j, t)) for t in ref_mm._refill_tags(tags, None, None, True)])
    store["codecv1"] = np.array(ref_pu.codecv1_to_frame2(), np.int64)
    meta["complement"] = {s: ref_pu.complement_seq(s) for s in ("ACGTN", "AACCGGTTRYKMBDHV", "acgt", "")}
    meta["base2code_dna"] = ref_pu.base2code_dna
    np.savez_compressed(os.path.join(HERE, "pipeline_golden.npz"), **store)
    json.dump(meta, open(os.path.join(HERE, "pipeline_golden.json"), "w"), indent=1, sort_keys=True)
    print("sites", n, "batches", batch_num)


def _j(x):
    if isinstance(x, (list, tuple)):
        return [_j(y) for y in x]
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (np.floating,)):
        return float(x)
    return x


if __name__ == "__main__":
    if sys.argv[1:] == ["variants"]:          # only the model-variant fixtures
        gen_forward_variants()
        sys.exit(0)
    gen_forward()
    gen_forward_variants()
    gen_extract_and_pipeline()
