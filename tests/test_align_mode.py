"""`call_mods --mode align`: read filters and the aligned-part site window (extract_features.py:272-304, 383-391) against the
REFERENCE's own extraction on duck-typed aligned reads (tests/golden/make_align_golden.py)."""
import argparse
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from ccsmeth_amd import bamio, bamnative
from ccsmeth_amd import call_mods as cm
from ccsmeth_amd import extract_features as ef
from ccsmeth_amd.utils import synth

GOLD = json.load(open(os.path.join(GOLDEN, "align_golden.json")))
READS = synth.synth_aligned_reads(GOLD["seed"], GOLD["n"])


def _args(over):
    a = cm.build_parser().parse_args(["-i", "x.bam", "-m", "m.ckpt", "-o", "o"])
    for k, v in over.items():
        setattr(a, k, v)
    if "mode" not in over:
        a.mode = "align"
    return a


def _write_bam(path):
    with bamio.BamWriter(path, "@HD\tVN:1.6\tSO:unknown\n@SQ\tSN:chr1\tLN:100000\n", [("chr1", 100000)]) as w:
        for name, flag, mapq, cigar, seq, fi, ri, fp, rp, fn, rn in READS:
            w.write(bamio.BamRecord(name, flag=flag, ref_id=0, pos=1000, mapq=mapq, cigar=cigar, seq=seq,
                                    tags=[("fi", "BC", fi), ("ri", "BC", ri), ("fp", "BC", fp), ("rp", "BC", rp), ("fn", "C", fn), ("rn", "C", rn)]))


@pytest.mark.parametrize("case", sorted(GOLD["cases"]))
def test_filters_and_window_reproduce_reference_sites(case, tmp_path):
    args = _args(GOLD["cases"][case]["args"])
    path = str(tmp_path / "a.bam")
    _write_bam(path)
    with bamnative.NativeBamReader(path, threads=2) as rd:
        b = rd.next_batch(100)
        mq, qs, qe, ident = bamnative.align_info(b)
        flag, length = b.flag.copy(), b.length.copy()
        b.close()
    for i, r in enumerate(READS):                       # native CIGAR summary == the Python one
        assert (int(qs[i]), int(qe[i])) == cm._cigar_align_info(r[3], len(r[4]))[:2] and mq[i] == r[2]
        assert abs(ident[i] - cm._cigar_align_info(r[3], len(r[4]))[2]) < 1e-15
    if args.mode == "align":
        skip, window = cm._align_skip_and_window(flag, mq, ident, qs, qe, length, args)
    else:
        skip, window = np.zeros(len(READS), bool), None
    for i, (name, fl, mapq, cigar, seq, fi, ri, fp, rp, fn, rn) in enumerate(READS):
        fwd = bamio.BamRecord(name, flag=fl, seq=seq).get_forward_sequence()
        arrs = None if skip[i] else ef.extract_read_arrays(fwd, fi, ri, fp, rp)
        locs = np.asarray([] if arrs is None else arrs["loc"], np.int64)
        if window is not None and len(locs):
            locs = locs[(locs >= window[0][i]) & (locs < window[1][i])]
        assert locs.tolist() == GOLD["cases"][case]["locs"][name], (case, name)


def test_site_window_filter_compacts_arrays():
    first = np.array([0, 3, 3, 7], np.int32)
    locs = np.array([5, 20, 40, 1, 2, 30, 31], np.int32)
    prob = np.arange(7, dtype=np.float32)
    f, l, p, t = cm._filter_sites_by_window(first, locs, prob, np.array([1, 0, 1], np.uint8), (np.array([10, 0, 50]), np.array([45, 9, 60])))
    assert f.tolist() == [0, 2, 2, 2] and l.tolist() == [20, 40] and p.tolist() == [1.0, 2.0] and t.tolist() == [1, 0, 0]
    with pytest.raises(ValueError):
        cm._check_scope(_args({"mode": "reference"}))


@pytest.mark.gpu
@pytest.mark.parametrize("io", ["native", "python"])
def test_call_mods_align_mode_end_to_end(io, tmp_path):
    """BAM -> modbam with --mode align: the tagged C positions of every read are the reference's kept sites."""
    import torch
    from collections import OrderedDict
    path = str(tmp_path / "a.bam")
    _write_bam(path)
    ckpt = str(tmp_path / "m.ckpt")
    torch.save(OrderedDict((k, torch.from_numpy(v)) for k, v in synth.synth_weights(5).items()), ckpt)
    for case in ("default", "keep_clipped", "mapq20_nosupp"):
        over = GOLD["cases"][case]["args"]
        argv = ["-i", path, "-m", ckpt, "-o", str(tmp_path / (case + io)), "--mode", "align", "--io", io, "--no_sort"]
        for k, v in over.items():
            argv += ["--" + k] + ([] if v is True else [str(v)])
        res = cm.call_mods(cm.build_parser().parse_args(argv), log=open(os.devnull, "w"))
        with bamio.BamReader(res["output"]) as rd:
            out = {o.query_name: o for o in rd}
        for name, fl, mapq, cigar, seq, *_ in READS:
            want = GOLD["cases"][case]["locs"][name]
            o = out[name]
            if not want:
                assert not o.has_tag("MM"), (case, name)
                continue
            fwd = o.get_forward_sequence()
            cs = np.flatnonzero(np.frombuffer(fwd.encode(), np.uint8) == ord("C"))
            deltas = [int(x) for x in o.get_tag("MM")[len("C+m?,"):-1].split(",")]
            got = cs[np.cumsum(np.array(deltas) + 1) - 1].tolist()
            assert got == want and len(o.get_tag("ML")) == len(want), (case, name)


# ---- `extract`: the feature table trainm reads, text-identical to the reference's writer --------------------------------
@pytest.mark.parametrize("case", sorted(GOLD["cases"]))
def test_extract_table_is_the_reference_text(case, tmp_path):
    import hashlib
    from ccsmeth_amd import extract_cli as ex
    path = str(tmp_path / "a.bam")
    _write_bam(path)
    over = GOLD["cases"][case]["args"]
    argv = ["-i", path, "-o", str(tmp_path / "f.tsv"), "--ref", path] + (["--mode", "align"] if "mode" not in over else [])
    for k, v in over.items():
        argv += ["--" + k] + ([] if v is True else [str(v)])
    res = ex.extract_hifireads_features(ex.build_parser().parse_args(argv), log=open(os.devnull, "w"))
    by_read = {}
    with open(res["output"]) as rf:
        for line in rf:
            by_read.setdefault(line.split("\t")[3], []).append(line)
    for name, gl in GOLD["cases"][case]["lines"].items():
        lines = by_read.get(name, [])
        assert [int(x.split("\t")[1]) for x in lines] == gl["pos"], (case, name)
        if lines:
            assert lines[0].rstrip("\n") == gl["first"], (case, name)
        assert hashlib.sha256("".join(lines).encode()).hexdigest() == gl["sha256"], (case, name)
    assert res["sites"] == sum(len(v) for v in GOLD["cases"][case]["locs"].values())


def test_extract_cli_flags_gzip_and_feeds_trainm_reader(tmp_path):
    import gzip
    from ccsmeth_amd import extract_cli as ex
    from ccsmeth_amd import trainm
    cli = json.load(open(os.path.join(GOLDEN, "cli_golden.json")))["extract"]
    ours = {a.dest: a for a in ex.build_parser()._actions if a.dest != "help"}
    for dest, g in cli.items():
        assert dest in ours and sorted(ours[dest].option_strings) == sorted(g["options"]) and ours[dest].default == g["default"], dest
    path = str(tmp_path / "a.bam")
    _write_bam(path)
    res = ex.extract_hifireads_features(ex.build_parser().parse_args(["-i", path, "--gzip", "--methy_label", "0"]), log=open(os.devnull, "w"))
    assert res["output"] == str(tmp_path / "a.features.tsv.gz")
    with gzip.open(res["output"], "rt") as rf, open(str(tmp_path / "plain.tsv"), "w") as wf:
        wf.write(rf.read())
    d = trainm.read_feature_file(str(tmp_path / "plain.tsv"))
    assert len(d["labels"]) == res["sites"] and set(d["labels"].tolist()) == {0} and d["kmer1"][:, 10].tolist() == [1] * res["sites"]
    for bad in (["--seq_len", "20"], ["--mode", "align"], ["--norm", "mad"], ["--is_map", "yes"], ["--motifs", "GATC"]):
        with pytest.raises(ValueError):
            ex.extract_hifireads_features(ex.build_parser().parse_args(["-i", path] + bad))
    with pytest.raises(IOError):
        ex.extract_hifireads_features(ex.build_parser().parse_args(["-i", str(tmp_path / "none.bam")]))
