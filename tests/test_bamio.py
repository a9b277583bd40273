"""BGZF/BAM reader-writer: specification-level checks and round trips (no htslib in this image to compare with)."""
import gzip
import os
import struct

import numpy as np
import pytest

from ccsmeth_amd import bamio


def _records(rng, n=7):
    recs = []
    for i in range(n):
        L = int(rng.integers(1, 400))
        seq = "".join(rng.choice(list("ACGTN"), size=L))
        tags = [("fi", "BC", rng.integers(0, 256, L).astype(np.uint8)), ("ri", "BC", rng.integers(0, 256, L).astype(np.uint8)),
                ("fp", "BC", rng.integers(0, 256, L).astype(np.uint8)), ("rp", "BC", rng.integers(0, 256, L).astype(np.uint8)),
                ("fn", "C", int(rng.integers(1, 200))), ("rn", "S", int(rng.integers(256, 1000))),
                ("sn", "Bf", rng.random(4).astype(np.float32)), ("rq", "f", 0.5), ("RG", "Z", "abc123"), ("zm", "i", -5 - i),
                ("XA", "A", "q")]
        aligned = i % 3 == 0
        recs.append(bamio.BamRecord("m64011/%d/ccs" % i, flag=(16 if i % 2 else 0) if aligned else 4,
                                    ref_id=0 if aligned else -1, pos=100 * i if aligned else -1, mapq=60 if aligned else 255,
                                    cigar=((0, L),) if aligned else (), seq=seq,
                                    qual=None if i % 2 else rng.integers(0, 94, L).astype(np.uint8), tags=tags))
    return recs


def test_roundtrip_and_bgzf_structure(tmp_path):
    rng = np.random.default_rng(1)
    recs = _records(rng, 40)
    path = str(tmp_path / "t.bam")
    hdr = "@HD\tVN:1.5\tSO:unknown\n@SQ\tSN:chr20\tLN:100000\n"
    with bamio.BamWriter(path, hdr, [("chr20", 100000)]) as w:
        for r in recs:
            w.write(r)
    raw = open(path, "rb").read()
    assert raw.endswith(bamio._BGZF_EOF) and raw[:4] == b"\x1f\x8b\x08\x04"
    plain = gzip.decompress(raw)                      # BGZF is a valid multi-member gzip stream
    assert plain[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", plain, 4)[0]
    assert plain[8:8 + l_text].decode() == hdr
    with bamio.BamReader(path) as rd:
        assert rd.header_text == hdr and rd.references == [("chr20", 100000)]
        got = list(rd)
    assert len(got) == len(recs)
    for a, b in zip(recs, got):
        assert (a.query_name, a.flag, a.ref_id, a.pos, a.mapq, a.cigar, a.seq) == (b.query_name, b.flag, b.ref_id, b.pos, b.mapq, b.cigar, b.seq)
        assert (a.qual is None) == (b.qual is None) and (a.qual is None or np.array_equal(a.qual, b.qual))
        assert [t[:2] for t in a.tags] == [t[:2] for t in b.tags]
        for (_, ty, va), (_, _, vb) in zip(a.tags, b.tags):
            if ty[0] == "B":
                assert np.array_equal(va, vb) and vb.dtype == np.dtype(bamio._B_DTYPES[ty[1]])
            elif ty == "f":
                assert abs(va - vb) < 1e-7
            else:
                assert va == vb
    assert got[1].is_reverse == bool(recs[1].flag & 16)


def test_large_file_spans_many_blocks(tmp_path):
    rng = np.random.default_rng(2)
    path = str(tmp_path / "big.bam")
    recs = []
    with bamio.BamWriter(path, "@HD\tVN:1.5\n", []) as w:
        for i in range(30):
            L = 15000
            r = bamio.BamRecord("r%d" % i, seq="".join(rng.choice(list("ACGT"), size=L)),
                                tags=[("fi", "BC", rng.integers(0, 256, L).astype(np.uint8)), ("fn", "C", 7)])
            recs.append(r)
            w.write(r)
    assert os.path.getsize(path) > 3 * 65536            # records straddle BGZF blocks
    with bamio.BamReader(path) as rd:
        got = list(rd)
    assert [g.seq for g in got] == [r.seq for r in recs]
    assert all(np.array_equal(g.get_tag("fi"), r.get_tag("fi")) for g, r in zip(got, recs))


def test_forward_sequence_and_helpers():
    r = bamio.BamRecord("x", flag=16, seq="AACGT")
    assert r.get_forward_sequence() == "ACGTT" and bamio.BamRecord("y", seq="AACGT").get_forward_sequence() == "AACGT"
    assert [bamio.int_tag_type(v) for v in (0, 255, 256, 65535, 65536, -1, -128, -129, -40000)] == list("CCSSIccsi")
    with pytest.raises(KeyError):
        r.get_tag("fi")
    assert bamio.add_pg_line("@HD\tVN:1.5", "0.5.0", "ccsmeth call_mods").splitlines()[-1] == \
        "@PG\tID:ccsmeth\tPN:ccsmeth\tVN:0.5.0\tCL:ccsmeth call_mods"


def test_corrupt_stream_is_rejected(tmp_path):
    path = str(tmp_path / "c.bam")
    with bamio.BamWriter(path, "@HD\tVN:1.5\n", []) as w:
        w.write(bamio.BamRecord("a", seq="ACGT"))
    raw = bytearray(open(path, "rb").read())
    raw[30] ^= 0xff
    open(path, "wb").write(raw)
    with pytest.raises((ValueError, Exception)):
        list(bamio.BamReader(path))
