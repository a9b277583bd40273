"""`call_freqb` (aligned modbam -> per-site frequencies): libccsm_bam's per-record projection + the host mirror of
call_mods_freq_bam.py, against the text the REFERENCE's own functions produced for the same synthetic modbam
(tests/golden/make_freqb_golden.py).  Count mode must be byte-identical; aggregate mode runs here on the NumPy oracle of the
aggregate model (the GPU model is compared in tests/test_gpu_parity.py)."""
import argparse
import gzip
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from ccsmeth_amd import bamio, bamnative
from ccsmeth_amd import call_mods_freq_bam as fb

BAM = os.path.join(GOLDEN, "freqb", "aligned.modbam.bam")
REF = os.path.join(GOLDEN, "freqb", "ref.fa")
with gzip.open(os.path.join(GOLDEN, "freqb_golden.json.gz"), "rt") as _f:
    CASES = json.load(_f)


def _args(case, out, **kw):
    a = fb.build_freqb_parser().parse_args(["--input_bam", BAM, "--ref", REF, "-o", out])
    for k, v in CASES[case]["args"].items():
        setattr(a, k, v)
    a.threads = 3
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def _read(path):
    if not os.path.exists(path):
        return ""
    with open(path) as rf:
        return rf.read()


def _outputs(prefix, a):
    ext = "bed" if a.bed else "freq.txt"
    return {w: _read("{}.{}.{}.{}".format(prefix, a.call_mode, w, ext)) for w in ("all", "hp1", "hp2")}


class OracleAggrModel:
    """new_region()/forward_raw() of AggrModel on the NumPy oracle + torch.randn replica (test-only stand-in)."""

    def __init__(self):
        from oracle import attbigru2s_oracle as orc
        from oracle.torch_randn_replica import Mt19937Stream, normals_from_raw
        self.orc = orc
        self.w = dict(np.load(os.path.join(GOLDEN, "aggr_ckpt_weights.npz")))
        s = Mt19937Stream(1234)
        s.raw(sum(v.size for v in self.w.values()))
        self.normals = normals_from_raw(s.raw(64 * 12000))
        self.pos = 0

    def new_region(self):
        self.pos = 0

    def forward_raw(self, refposes, histos, only_close=False):
        out, self.pos = self.orc.cal_modfreq_in_aggregate_mode(np.asarray(refposes), np.asarray(histos), self.w, self.normals, self.pos,
                                                               only_close=only_close)
        return out


COUNT_CASES = sorted(k for k in CASES if k.startswith("count"))
AGGR_CASES = sorted(k for k in CASES if k.startswith("aggregate"))


def test_parser_defaults_match_reference_cli():
    a = fb.build_freqb_parser().parse_args(["--input_bam", "x.bam", "--ref", "r.fa", "-o", "o"])
    want = dict(threads=5, contigs=None, chunk_len=500000, bed=False, sort=False, gzip=False, modtype="5mC", call_mode="count",
                prob_cf=0.0, no_amb_cov=False, hap_tag="HP", mapq=1, identity=0.0, no_supplementary=False, motifs="CG", mod_loc=0,
                no_comb=False, refsites_only=False, refsites_all=False, no_hap=False, base_clip=0, aggre_model=None,
                model_type="attbigru", seq_len=11, class_num=1, layer_rnn=1, hid_rnn=32, bin_size=20, cov_cf=4, only_close=False,
                discrete=False, tseed=1234)                                  # ccsmeth.py:460-556
    for k, v in want.items():
        assert getattr(a, k) == v, k


@pytest.mark.parametrize("case", COUNT_CASES)
def test_count_mode_text_identical_to_reference(case, tmp_path):
    a = _args(case, str(tmp_path / "o"))
    assert [list(c) for c in fb._get_reference_chunks(fb.read_fasta(REF)[1], a.contigs, a.chunk_len, a.motifs)] == CASES[case]["chunks"]
    fb.call_mods_frequency_from_bamfile(a, log=open(os.devnull, "w"))
    got = _outputs(str(tmp_path / "o"), a)
    for w in ("all", "hp1", "hp2"):
        assert got[w] == CASES[case][w], (case, w)
        if not CASES[case][w]:
            assert not os.path.exists("{}.count.{}.{}".format(tmp_path / "o", w, "bed" if a.bed else "freq.txt"))   # empty files removed


def _close(got, want, bed):
    g, w = got.splitlines(), want.splitlines()
    assert len(g) == len(w)
    for a, b in zip(g, w):
        fa, fb_ = a.split("\t"), b.split("\t")
        if bed:
            assert fa[:10] == fb_[:10] and abs(int(fa[10]) - int(fb_[10])) <= 1, (a, b)
        else:
            assert fa[:6] == fb_[:6] and fa[8] == fb_[8] and fa[10] == fb_[10], (a, b)
            assert abs(float(fa[9]) - float(fb_[9])) <= 1.01e-4 and abs(float(fa[6]) - float(fb_[6])) <= 1.0 + 1e-6, (a, b)
    return sum(x != y for x, y in zip(g, w))


@pytest.mark.parametrize("case", AGGR_CASES)
def test_aggregate_mode_on_oracle_model_matches_reference(case, tmp_path):
    a = _args(case, str(tmp_path / "o"))
    fb.call_mods_frequency_from_bamfile(a, log=open(os.devnull, "w"), model=OracleAggrModel())
    got = _outputs(str(tmp_path / "o"), a)
    diff = total = 0
    for w in ("all", "hp1", "hp2"):
        diff += _close(got[w], CASES[case][w], a.bed)
        total += len(got[w].splitlines())
    assert diff <= 0.002 * total          # the model's 6th decimal may round the other way; everything else is exact


def test_sort_and_gzip(tmp_path):
    a = _args("count_no_comb", str(tmp_path / "o"), sort=True)
    fb.call_mods_frequency_from_bamfile(a, log=open(os.devnull, "w"))
    lines = _read(str(tmp_path / "o") + ".count.all.freq.txt").splitlines()
    keys = [(ln.split("\t")[0], int(ln.split("\t")[1])) for ln in lines]
    assert keys == sorted(keys) and sorted(lines) == sorted(CASES["count_no_comb"]["all"].splitlines())
    b = _args("count_no_comb", str(tmp_path / "g"), gzip=True)
    fb.call_mods_frequency_from_bamfile(b, log=open(os.devnull, "w"))
    p = str(tmp_path / "g") + ".count.all.freq.txt"
    assert not os.path.exists(p)
    with gzip.open(p + ".gz", "rt") as rf:
        text = rf.read()
    assert text.splitlines() == lines                                # --gzip implies the sort, like the reference
    with open(p + ".gz", "rb") as rf:
        assert list(bamio.bgzf_blocks(rf))                       # BGZF framing, not plain gzip
    # the tabix index: parse it and use it for region queries against a scan of the text
    import struct
    import zlib
    with open(p + ".gz.tbi", "rb") as rf:
        tbi = b"".join(bamio.bgzf_blocks(rf))
    assert tbi[:4] == b"TBI\x01"
    n_ref, fmt, cs, cb, ce, meta, skip, l_nm = struct.unpack_from("<8i", tbi, 4)
    assert (fmt, cs, cb, ce, meta, skip) == (0x10000, 1, 2, 3, ord("#"), 0)
    names = tbi[36:36 + l_nm].split(b"\x00")[:-1]
    assert [n.decode() for n in names] == sorted(set(ln.split("\t")[0] for ln in lines))
    off = 36 + l_nm
    raw = open(p + ".gz", "rb").read()

    def read_from(v0, v1):                                           # decompress from virtual offset v0 up to v1
        out, c = b"", v0 >> 16
        while c <= (v1 >> 16) and c < len(raw) - 28:
            bsize = struct.unpack_from("<H", raw, c + 16)[0] + 1
            payload = zlib.decompress(raw[c + 18:c + bsize - 8], -15)
            lo = (v0 & 0xffff) if c == (v0 >> 16) else 0
            hi = (v1 & 0xffff) if c == (v1 >> 16) else len(payload)
            out += payload[lo:hi]
            c += bsize
        return out
    rng = np.random.default_rng(1)
    for name in names:
        (n_bin,) = struct.unpack_from("<i", tbi, off); off += 4
        bins = {}
        for _ in range(n_bin):
            b_, n_chunk = struct.unpack_from("<Ii", tbi, off); off += 8
            bins[b_] = [struct.unpack_from("<QQ", tbi, off + 16 * k) for k in range(n_chunk)]
            off += 16 * n_chunk
        (n_intv,) = struct.unpack_from("<i", tbi, off); off += 4
        off += 8 * n_intv
        mine = [ln for ln in lines if ln.split("\t")[0] == name.decode()]
        assert bins.pop(37450)[1][0] == len(mine)
        for _ in range(10):
            beg = int(rng.integers(0, 5000)); end = beg + int(rng.integers(1, 3000))
            want = [ln for ln in mine if int(ln.split("\t")[1]) < end and int(ln.split("\t")[2]) > beg]
            cand = []
            for b_ in fb._reg2bins(beg, end):
                for c0, c1 in bins.get(b_, []):
                    cand += read_from(c0, c1).decode().splitlines()
            assert set(want) <= set(cand), (name, beg, end)
    assert off + 0 <= len(tbi)


def test_errors(tmp_path):
    a = _args("count_default", str(tmp_path / "o"))
    a.input_bam = str(tmp_path / "x.sam")
    with pytest.raises(ValueError, match="not a bam file"):
        fb.call_mods_frequency_from_bamfile(a)
    a.input_bam = str(tmp_path / "missing.bam")
    with pytest.raises(ValueError, match="does not exist"):
        fb.call_mods_frequency_from_bamfile(a)
    a = _args("count_default", str(tmp_path / "o"), ref=str(tmp_path / "none.fa"))
    with pytest.raises(ValueError, match="--ref does not exist"):
        fb.call_mods_frequency_from_bamfile(a)
    a = _args("aggregate_default", str(tmp_path / "o"), aggre_model=None)
    with pytest.raises(ValueError, match="--aggre_model is not set right"):
        fb.call_mods_frequency_from_bamfile(a)
    a = _args("count_default", str(tmp_path / "o"), contigs="nope")
    with pytest.raises(ValueError, match="not in --ref"):
        fb.call_mods_frequency_from_bamfile(a)
    # a modbam whose contigs come back after a later one (e.g. call_mods --no_sort output of shuffled reads) is rejected whatever the
    # record batching: here the whole file is one batch, and the out-of-place record sits in the middle of it
    with bamio.BamReader(BAM) as rd:
        hdr, refs, recs = rd.header_text, rd.references, list(rd)
    tids = sorted({r.ref_id for r in recs if r.ref_id >= 0})
    if len(tids) >= 2:
        first = [r for r in recs if r.ref_id == tids[0]]
        second = [r for r in recs if r.ref_id == tids[1]]
        shuffled = str(tmp_path / "shuffled.bam")
        with bamio.BamWriter(shuffled, hdr, refs) as w:
            for r in first[:-1] + second[:3] + first[-1:] + second[3:]:
                w.write(r)
        a = _args("count_default", str(tmp_path / "o"), input_bam=shuffled)
        with pytest.raises(ValueError, match="not coordinate-sorted"):
            fb.call_mods_frequency_from_bamfile(a, log=open(os.devnull, "w"))


# ---- the per-record projection against a pure-Python restatement (covers what the goldens' generator cannot: several
#      modification groups in one MM tag, as htslib's parser - the reference's first choice - reads them)
def _py_modcalls(rec, tid_ok=True, mapq=1, base_clip=0):
    if rec.flag & (0x4 | 0x100 | 0x400) or rec.mapq < mapq:
        return []
    try:
        mm, ml = rec.get_tag("MM"), rec.get_tag("ML")
    except KeyError:
        return []
    fwd = rec.get_forward_sequence()
    off, calls = 0, None
    for grp in mm.split(";"):
        if not grp:
            continue
        head = grp.split(",")[0]
        deltas = [int(x) for x in grp.split(",")[1:]]
        codes = head[2:].rstrip("?.")
        if head.startswith("C+m") and calls is None:
            cs = [i for i, c in enumerate(fwd) if c == "C"]
            ords = np.cumsum(np.array(deltas) + 1) - 1
            calls = [(cs[o], int(ml[off + k])) for k, o in enumerate(ords)]
        off += len(deltas) * max(1, len(codes))
    if calls is None or off != len(ml):
        return []
    L = len(fwd)
    qmap = {(L - 1 - p if rec.is_reverse else p): v for p, v in calls}
    pairs, q, r = [], 0, rec.pos
    for op, ln in rec.cigar:
        if op in (0, 7, 8):
            pairs += [(q + i, r + i) for i in range(ln)]
            q += ln
            r += ln
        elif op in (1, 4):
            q += ln
        elif op in (2, 3):
            r += ln
    if base_clip:
        pairs = pairs[base_clip:-base_clip]
    return [(rp, int(rec.is_reverse), qmap[qp]) for qp, rp in pairs if qp in qmap]


def test_modcalls_multi_group_mm_and_hap_types(tmp_path):
    rng = np.random.default_rng(5)
    ref = "".join(rng.choice(list("ACGT"), size=3000))
    recs = []
    for i in range(40):
        st = int(rng.integers(0, 2000))
        seq = ref[st:st + 600]
        rev = i % 2 == 1
        fwd = bamio.BamRecord("r", flag=16 if rev else 0, seq=seq).get_forward_sequence()
        cs = [k for k, c in enumerate(fwd) if c == "C"]
        pick = sorted(rng.choice(len(cs), size=20, replace=False).tolist())
        d_m = [pick[0]] + [pick[k] - pick[k - 1] - 1 for k in range(1, len(pick))]
        pick_h = sorted(rng.choice(len(cs), size=7, replace=False).tolist())
        d_h = [pick_h[0]] + [pick_h[k] - pick_h[k - 1] - 1 for k in range(1, len(pick_h))]
        ml_h = rng.integers(0, 256, size=7).astype(np.uint8)
        ml_m = rng.integers(0, 256, size=20).astype(np.uint8)
        if i % 3 == 0:
            mm = "C+h?," + ",".join(map(str, d_h)) + ";C+m?," + ",".join(map(str, d_m)) + ";"
            ml = np.concatenate([ml_h, ml_m])
        elif i % 3 == 1:
            mm = "C+m," + ",".join(map(str, d_m)) + ";A+a?;C+h," + ",".join(map(str, d_h)) + ";"
            ml = np.concatenate([ml_m, ml_h])
        else:
            mm = "C+m.," + ",".join(map(str, d_m)) + ";"
            ml = ml_m
        hp = [("HP", "i", 1), ("HP", "C", 2), ("HP", "Z", "2"), ("HP", "A", "1"), ("HP", "f", 2.0), ("HP", "Z", "1x"), ("HP", "i", 0)][i % 7]
        recs.append(bamio.BamRecord("r%d" % i, flag=16 if rev else 0, ref_id=0, pos=st, mapq=30, cigar=[(4, 5), (0, 295), (2, 3), (0, 300)],
                                    seq="ACGTA" + seq[:295] + seq[298:598], tags=[hp, ("MM", "Z", mm), ("ML", "BC", ml)]))
        # rebuild tags over the record's real sequence
        r = recs[-1]
        fwd = r.get_forward_sequence()
        if max(pick + pick_h) >= fwd.count("C"):
            recs.pop()
    path = str(tmp_path / "m.bam")
    with bamio.BamWriter(path, "@HD\tVN:1.6\n@SQ\tSN:c\tLN:3000\n", [("c", 3000)]) as bw:
        for r in recs:
            bw.write(r)
    with bamnative.NativeBamReader(path, threads=2) as rd:
        batch = rd.next_batch(1000)
        for clip in (0, 9):
            tid, pos, strand, ml, hap, seen, used = bamnative.modcalls_of_batch(batch, mapq=1, base_clip=clip, threads=3)
            want = [row for r in recs for row in _py_modcalls(r, base_clip=clip)]
            assert seen == used == len(recs)
            assert list(zip(pos.tolist(), strand.tolist(), ml.tolist())) == want and len(want) > 300
        want_hap = {"i1": 1, "C2": 2, "Z2": 2, "A1": 1, "f2.0": 2, "Z1x": 0, "i0": 0}
        per_read_hap = []
        k = 0
        for r in recs:
            n = len(_py_modcalls(r, base_clip=9))
            per_read_hap.append((r.tags[0][1] + str(r.tags[0][2]), set(hap[k:k + n].tolist())))
            k += n
        for key, hs in per_read_hap:
            assert hs <= {want_hap[key]}, (key, hs)
        with pytest.raises(IOError, match="site masks"):
            bamnative.modcalls_of_batch(batch, refsites_all=True)
        batch.close()


def test_motif_site_mask_is_per_region():
    seq = "ACGCGTTCGACCGG"
    regions = [(0, 4), (4, 9), (9, 14)]
    mask = fb._motif_site_mask(seq, regions, ["CG"], 0)
    fwd, rev = [], []
    for s, e in regions:                                          # call_mods_freq_bam.py:473-479, restated naively
        sl = seq[s:e]
        fwd += [s + i for i in range(len(sl) - 1) if sl[i:i + 2] == "CG"]
        rc = fb.complement_seq(sl)
        rev += [e - 1 - i for i in range(len(rc) - 1) if rc[i:i + 2] == "CG"]
    assert np.flatnonzero(mask & 1).tolist() == sorted(fwd)
    assert np.flatnonzero(mask & 2).tolist() == sorted(rev)
    assert fb.get_motif_seqs("CHG,CG") == ["CAG", "CCG", "CTG", "CG"]
    assert fb.complement_seq("ACGTNRX") == "NYNACGT"
