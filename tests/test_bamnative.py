"""libccsm_bam (native threaded BGZF/BAM reader + modbam writer) against the pure-Python bamio / _bam2modbam mirrors, which
are themselves pinned to the reference's outputs (tests/test_host_mirror.py, tests/test_bamio.py).  CPU only."""
import os

import numpy as np
import pytest

from ccsmeth_amd import _bam2modbam as mmod
from ccsmeth_amd import bamio, bamnative
from ccsmeth_amd import extract_features as ef


def _make_bam(path, rng, n_reads=40):
    recs = []
    with bamio.BamWriter(path, "@HD\tVN:1.5\tSO:unknown\n@RG\tID:x\n", [("chr1", 1000), ("chrUn_2", 77)]) as w:
        for i in range(n_reads):
            L = int(rng.choice([0, 1, 19, 21, 22, 23, 64, 257, 1000, 4097, 70001][: 11 if i % 13 == 0 else 9]))
            seq = "".join(rng.choice(list("ACGTN"), size=L, p=[0.27, 0.22, 0.22, 0.27, 0.02]))
            kin = lambda: rng.integers(0, 256, L).astype(np.uint8)  # noqa: E731
            tags = [("np", "C", 12), ("fi", "BC", kin()), ("zm", "i", 100000 + i), ("fp", "BC", kin()), ("ri", "BC", kin()),
                    ("rp", "BC", kin()), ("fn", ["C", "c", "S", "s", "I", "i"][i % 6], int(rng.integers(3, 100))),
                    ("rn", "C", int(rng.integers(3, 100))), ("sn", "Bf", rng.random(4).astype(np.float32)), ("RG", "Z", "x"),
                    ("MM", "Z", "C+m,1;"), ("ML", "BC", np.array([7], np.uint8)), ("rq", "f", 0.999)]
            if i % 7 == 3:
                tags[1] = ("fi", "BC", kin()[: max(L - 1, 0)])       # wrong length -> unusable
            if i % 11 == 5:
                tags = [t for t in tags if t[0] != "rp"]             # missing tag -> unusable
            if i % 9 == 4:
                tags[4] = ("ri", "BS", rng.integers(0, 256, L).astype(np.uint16))   # wrong element type -> unusable
            flag = [4, 0, 16, 4 | 16][i % 4]
            r = bamio.BamRecord("m64/%d/ccs" % i, flag=flag, ref_id=0 if not flag & 4 else -1, pos=10 * i if not flag & 4 else -1,
                                mapq=60, cigar=((0, L),) if L and not flag & 4 else (), seq=seq,
                                qual=None if i % 5 == 0 else rng.integers(0, 60, L).astype(np.uint8), tags=tags)
            recs.append(r)
            w.write(r)
    return recs


def _usable(r):
    L = len(r.seq)
    try:
        arrs = [r.get_tag(t) for t in ("fi", "ri", "fp", "rp")]
    except KeyError:
        return False
    return L > 0 and all(isinstance(a, np.ndarray) and a.dtype == np.uint8 and len(a) == L for a in arrs)


@pytest.mark.parametrize("threads,chunk", [(1, 7), (4, 1000), (3, 1)])
def test_native_reader_matches_python_reader(tmp_path, threads, chunk):
    rng = np.random.default_rng(11)
    path = str(tmp_path / "in.bam")
    recs = _make_bam(path, rng)
    with bamnative.NativeBamReader(path, threads=threads) as rd:
        assert rd.header_text == "@HD\tVN:1.5\tSO:unknown\n@RG\tID:x\n" and rd.n_ref == 2
        i = 0
        while True:
            b = rd.next_batch(chunk)
            if b is None:
                break
            assert 0 < b.n_reads <= chunk
            for k in range(b.n_reads):
                r = recs[i]
                assert b.flag[k] == r.flag
                raw = bytes(b.records[b.rec_offset[k]:b.rec_offset[k + 1]])
                assert int.from_bytes(raw[:4], "little") == len(raw) - 4
                if _usable(r):
                    L = len(r.seq)
                    o = int(b.offset[k])
                    assert b.length[k] == L
                    assert bytes(b.seq[o:o + L]).decode() == r.get_forward_sequence()
                    for name, arr in (("fi", b.fi), ("ri", b.ri), ("fp", b.fp), ("rp", b.rp)):
                        assert np.array_equal(arr[o:o + L], r.get_tag(name))
                    assert b.fn[k] == r.get_tag("fn") and b.rn[k] == r.get_tag("rn")
                    assert b.n_sites[k] == ef.count_kept_sites(np.frombuffer(r.get_forward_sequence().encode(), np.uint8))
                else:
                    assert b.length[k] == 0 and b.n_sites[k] == 0
                i += 1
            b.close()
        assert i == len(recs)


def test_native_writer_matches_python_refill(tmp_path):
    """Records written by the native writer (tag refill + MM/ML) parse back, with the Python reader, to exactly what the
    Python mirrors of _refill_tags / _convert_locs_to_mmtag / _convert_probs_to_mltag produce."""
    rng = np.random.default_rng(5)
    inp, outp = str(tmp_path / "in.bam"), str(tmp_path / "out.bam")
    recs = _make_bam(inp, rng, n_reads=30)
    expected = []
    with bamnative.NativeBamReader(inp, threads=2) as rd, \
            bamnative.NativeBamWriter(outp, rd.header_text + "@PG\tID:t\n", rd.raw_refs, rd.n_ref, threads=3, level=4) as wr:
        base = 0
        n_tagged_total = 0
        while True:
            b = rd.next_batch(8)
            if b is None:
                break
            first = np.zeros(b.n_reads + 1, np.int32)
            locs, probs, tagged = [], [], np.zeros(b.n_reads, np.uint8)
            for k in range(b.n_reads):
                r = recs[base + k]
                lk, pk = [], []
                if b.length[k] > 0 and b.n_sites[k] > 0:
                    fwd = r.get_forward_sequence()
                    cs = [j for j, c in enumerate(fwd) if c == "C"]
                    lk = sorted(rng.choice(cs, size=min(len(cs), int(rng.integers(1, 9))), replace=False).tolist())
                    pk = np.round(rng.random(len(lk)), 6).astype(np.float32)
                    pk[rng.random(len(lk)) < 0.1] = 1.0
                    tagged[k] = 1
                    if (base + k) % 5 == 2:
                        lk[-1] = next(j for j, c in enumerate(fwd) if c != "C")   # not a C -> the read stays untagged
                        lk = sorted(lk)
                locs += lk
                probs += list(pk)
                first[k + 1] = first[k] + len(lk)
                old = [(t, v) for t, _, v in r.tags]
                try:
                    mm = mmod._convert_locs_to_mmtag(lk, r.get_forward_sequence()) if tagged[k] else None
                except AssertionError:
                    mm = None
                ml = mmod._convert_probs_to_mltag(list(pk)) if mm is not None else None
                expected.append((r, mmod._refill_tags(old, mm, ml, rm_pulse=(base % 2 == 0)), mm is not None))
            n_tagged_total += wr.write_batch(b, first, np.array(locs, np.int32), np.array(probs, np.float32), tagged,
                                             rm_pulse=(base % 2 == 0))
            base += b.n_reads
            b.close()
    assert n_tagged_total == sum(1 for _, _, t in expected if t) and n_tagged_total > 3
    with bamio.BamReader(outp) as rd:
        assert rd.header_text.endswith("@PG\tID:t\n") and rd.references == [("chr1", 1000), ("chrUn_2", 77)]
        out = list(rd)
    assert len(out) == len(expected)
    for o, (r, tags, _) in zip(out, expected):
        assert (o.query_name, o.flag, o.ref_id, o.pos, o.mapq, o.cigar, o.seq) == (r.query_name, r.flag, r.ref_id, r.pos, r.mapq, r.cigar, r.seq)
        assert (o.qual is None and r.qual is None) or np.array_equal(o.qual, r.qual)
        assert [t for t, _, _ in o.tags] == [t for t, _ in tags]
        for (t, typ, v), (_, ev) in zip(o.tags, tags):
            if isinstance(v, np.ndarray) or isinstance(ev, (list, np.ndarray)):
                assert np.array_equal(np.asarray(v), np.asarray(ev)), t
            elif isinstance(v, float):
                assert abs(v - ev) < 1e-6
            else:
                assert v == ev, t


def test_native_reader_errors(tmp_path):
    p = str(tmp_path / "junk.bam")
    open(p, "wb").write(b"this is not a bam file at all, not even gzip")
    with pytest.raises(IOError):
        bamnative.NativeBamReader(p)
    with pytest.raises(IOError):
        bamnative.NativeBamReader(str(tmp_path / "missing.bam"))
    # truncated file: valid header, record cut in the middle of a block stream
    good = str(tmp_path / "good.bam")
    _make_bam(good, np.random.default_rng(1), n_reads=6)
    data = open(good, "rb").read()
    open(p, "wb").write(data[: len(data) // 2])
    with pytest.raises(IOError):
        with bamnative.NativeBamReader(p) as rd:
            while rd.next_batch(2) is not None:
                pass


def test_block_aligned_runs_stitch_back_in_order(tmp_path):
    """The multi-GPU merge: batches written by different writers as block-aligned runs (ccsm_bam_writer_flush) and stitched
    back round-robin give the records of the input, in input order."""
    rng = np.random.default_rng(3)
    inp = str(tmp_path / "in.bam")
    recs = _make_bam(inp, rng, n_reads=26)
    world, chunk = 3, 4
    parts = [str(tmp_path / ("part%d.bam" % r)) for r in range(world)]
    runs, header_end = [], None
    with bamnative.NativeBamReader(inp, threads=2) as rd:
        writers = [bamnative.NativeBamWriter(p, rd.header_text, rd.raw_refs, rd.n_ref, threads=2) for p in parts]
        ends = [w.flush() for w in writers]
        header_end = ends[0]
        bi = 0
        while True:
            b = rd.next_batch(chunk)
            if b is None:
                break
            r = bi % world
            writers[r].write_batch(b, rm_pulse=False)
            new_end = writers[r].flush()
            runs.append((parts[r], ends[r], new_end))
            ends[r] = new_end
            b.close()
            bi += 1
        for w in writers:
            w.close()
    out = str(tmp_path / "merged.bam")
    bamnative.stitch_runs(out, parts[0], header_end, runs)
    with bamio.BamReader(out) as rd:
        got = list(rd)
    assert [g.query_name for g in got] == [r.query_name for r in recs]
    for g, r in zip(got, recs):
        assert g.seq == r.seq and [t for t, _, _ in g.tags] == [t for t, _, _ in r.tags if t not in ("MM", "ML")]


def test_native_reader_survives_corrupted_records(tmp_path):
    """Random byte corruption of the (re-compressed) record stream: the native reader / writer either work or raise IOError;
    run in a child process so that a memory fault would be seen as a non-zero exit instead of killing the test session."""
    import subprocess
    import sys
    from conftest import ROOT
    script = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from ccsmeth_amd import bamio, bamnative
from test_bamnative import _make_bam
rng = np.random.default_rng(4)
path, mut, outp = %r, %r, %r
_make_bam(path, rng, n_reads=10)
raw = b"".join(bamio.bgzf_blocks(open(path, "rb")))
ok = err = 0
for trial in range(80):
    b = bytearray(raw)
    for _ in range(int(rng.integers(1, 6))):
        b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
    with open(mut, "wb") as f:
        for i in range(0, len(b), 0xff00):
            f.write(bamio.bgzf_compress_block(bytes(b[i:i + 0xff00]), 1))
        f.write(bamio._BGZF_EOF)
    try:
        with bamnative.NativeBamReader(mut, threads=2) as rd:
            while True:
                bt = rd.next_batch(4)
                if bt is None:
                    break
                _ = int(bt.records.sum()) + (int(bt.seq.sum()) if bt.total_bases else 0)
                w = bamnative.NativeBamWriter(outp, rd.header_text, rd.raw_refs, rd.n_ref, threads=1, level=1)
                w.write_batch(bt, rm_pulse=True)
                w.close()
                bt.close()
        ok += 1
    except IOError:
        err += 1
print("ok", ok, "ioerror", err)
assert ok + err == 80 and ok > 0
''' % (ROOT, os.path.join(ROOT, "tests"), str(tmp_path / "in.bam"), str(tmp_path / "mut.bam"), str(tmp_path / "out.bam"))
    res = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr


# ---- coordinate sort + BAI index (the reference's pysam.sort / pysam.index post-processing) -------------------------------
def _bai_parse(path):
    import struct
    b = open(path, "rb").read()
    assert b[:4] == b"BAI\x01"
    off = 4
    (n_ref,) = struct.unpack_from("<i", b, off); off += 4
    refs = []
    for _ in range(n_ref):
        (n_bin,) = struct.unpack_from("<i", b, off); off += 4
        bins = {}
        for _ in range(n_bin):
            bin_, n_chunk = struct.unpack_from("<Ii", b, off); off += 8
            bins[bin_] = [struct.unpack_from("<QQ", b, off + 16 * k) for k in range(n_chunk)]
            off += 16 * n_chunk
        (n_intv,) = struct.unpack_from("<i", b, off); off += 4
        lin = list(struct.unpack_from("<%dQ" % n_intv, b, off)); off += 8 * n_intv
        refs.append((bins, lin))
    (n_no_coor,) = struct.unpack_from("<Q", b, off); off += 8
    assert off == len(b)
    return refs, n_no_coor


def _record_voffsets(path):
    """(virtual offset, ref id, pos, end, flag) of every record, from an independent walk over the BGZF blocks."""
    import struct
    blocks, data = [], bytearray()
    with open(path, "rb") as fh:
        while True:
            c0 = fh.tell()
            head = fh.read(18)
            if len(head) < 18:
                break
            bsize = struct.unpack_from("<H", head, 16)[0] + 1
            fh.seek(c0)
            raw = fh.read(bsize)
            import zlib
            payload = zlib.decompress(raw[18:-8], -15)
            blocks.append((len(data), c0, len(payload)))
            data += payload
    def voff(p):
        for a, c, n in blocks:
            if a <= p < a + n:
                return (c << 16) | (p - a)
        a, c, n = blocks[-1]
        return (c << 16) | n
    l_text = struct.unpack_from("<i", data, 4)[0]
    off = 8 + l_text
    n_ref = struct.unpack_from("<i", data, off)[0]; off += 4
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", data, off)[0]; off += 8 + ln
    recs = []
    while off < len(data):
        bs = struct.unpack_from("<i", data, off)[0]
        tid, pos, l_name, _mq, _bin, n_cig, flag = struct.unpack_from("<iiBBHHH", data, off + 4)
        cig = np.frombuffer(bytes(data[off + 36 + l_name:off + 36 + l_name + 4 * n_cig]), "<u4")
        reflen = int(sum(int(c >> 4) for c in cig if int(c & 15) in (0, 2, 3, 7, 8)))
        recs.append((voff(off), tid, pos, pos + (reflen if reflen and not flag & 4 else 1), flag))
        off += 4 + bs
    return recs


def _reg2bins(beg, end):
    end -= 1
    out = [0]
    for shift, base in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
        out += list(range(base + (beg >> shift), base + (end >> shift) + 1))
    return out


def _aligned_records(rng, n, n_ref=3, ref_len=300000):
    recs = []
    for i in range(n):
        tid = int(rng.integers(0, n_ref))
        pos = int(rng.integers(0, ref_len - 20000))
        ln = int(rng.integers(200, 18000))
        flag = int(rng.choice([0, 16, 0x800, 0x100 | 16]))
        cigar = [(4, 3), (0, ln // 2), (2, 5), (1, 2), (0, ln - ln // 2)]
        seq = "".join(rng.choice(list("ACGT"), size=3 + ln + 2))
        recs.append(bamio.BamRecord("r%d" % i, flag=flag, ref_id=tid, pos=pos, mapq=30, cigar=cigar, seq=seq, tags=[("NM", "i", i)]))
    recs.append(bamio.BamRecord("placed_unmapped", flag=4, ref_id=1, pos=500, seq="ACGT"))
    for i in range(5):
        recs.append(bamio.BamRecord("u%d" % i, flag=4, seq="ACGTACGT"))
    return recs


def test_sort_and_index_aligned_bam(tmp_path):
    rng = np.random.default_rng(3)
    recs = _aligned_records(rng, 400)
    order = rng.permutation(len(recs))
    path = str(tmp_path / "a.bam")
    refs = [("c%d" % i, 300000) for i in range(3)]
    with bamio.BamWriter(path, "@HD\tVN:1.5\tSO:unsorted\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs) + "@CO\tx\n", refs, level=1) as w:
        for i in order:
            w.write(recs[i])
    ok, n = bamnative.index_build(path, threads=3)
    assert not ok and n == len(recs) and not os.path.exists(path + ".bai")
    assert bamnative.sort_and_index(path, threads=3) is True
    with bamio.BamReader(path) as rd:
        assert rd.header_text.startswith("@HD\tVN:1.5\tSO:coordinate\n@SQ") and rd.header_text.endswith("@CO\tx\n")
        got = list(rd)
    key = lambda r: ((r.ref_id if r.ref_id >= 0 else 1 << 31), r.pos + 1, 1 if r.flag & 16 else 0)  # noqa: E731
    want = sorted([recs[i] for i in order], key=key)                 # Python's sort is stable, like samtools'
    assert [r.query_name for r in got] == [r.query_name for r in want]
    assert all(a.cigar == b.cigar and a.seq == b.seq and a.tags[0][2] == b.tags[0][2] for a, b in zip(got[:50], want[:50]))
    assert bamnative.sort_and_index(path, threads=2) is False       # already sorted: index only
    # the index finds every overlapping record: SAM spec 5.3 query, checked against an independent walk of the file
    idx, n_no_coor = _bai_parse(path + ".bai")
    allrec = _record_voffsets(path)
    assert n_no_coor == 5 and len(idx) == 3 and len(allrec) == len(recs)
    for tid in range(3):
        bins, lin = idx[tid]
        meta = bins.pop(37450)
        mine = [r for r in allrec if r[1] == tid]
        assert meta[0] == (mine[0][0], [r for r in allrec][allrec.index(mine[-1]) + 1][0] if allrec.index(mine[-1]) + 1 < len(allrec) else meta[0][1])
        assert meta[1] == (sum(1 for r in mine if not r[4] & 4), sum(1 for r in mine if r[4] & 4))
        for _ in range(40):
            beg = int(rng.integers(0, 299000)); end = beg + int(rng.integers(1, 40000))
            overl = [r[0] for r in mine if r[2] < end and r[3] > beg]
            chunks = [c for b in _reg2bins(beg, end) for c in bins.get(b, [])]
            min_off = lin[beg >> 14] if (beg >> 14) < len(lin) else (lin[-1] if lin else 0)
            for v in overl:
                assert any(c0 <= v < c1 for c0, c1 in chunks), (tid, beg, end, v)
                assert v >= min_off


def test_sort_spills_runs_and_merges_to_the_same_file(tmp_path):
    """A sort memory limit far below the input: sorted runs go to temporary files and are merged; the result is record for record
    what the in-memory sort gives (stable: equal keys keep their input order across runs), and no temporary file is left."""
    rng = np.random.default_rng(8)
    recs = _aligned_records(rng, 600)
    for i in range(0, len(recs) - 6, 7):                              # runs of equal keys across the file
        recs[i].ref_id, recs[i].pos, recs[i].flag = 1, 4242, 0
    order = rng.permutation(len(recs))
    refs = [("c%d" % i, 300000) for i in range(3)]
    paths = [str(tmp_path / ("s%d.bam" % k)) for k in range(2)]
    for p in paths:
        with bamio.BamWriter(p, "@HD\tVN:1.5\tSO:unsorted\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs), refs, level=1) as w:
            for i in order:
                w.write(recs[i])
    assert bamnative.sort_and_index(paths[0], threads=2) is True
    assert bamnative.sort_and_index(paths[1], threads=2, max_bytes=200000) is True      # ~6 MB of records: dozens of runs
    with bamio.BamReader(paths[0]) as a, bamio.BamReader(paths[1]) as b:
        assert a.header_text == b.header_text
        ra, rb = list(a), list(b)
    assert len(ra) == len(rb) == len(recs)
    assert [(x.query_name, x.ref_id, x.pos, x.flag) for x in ra] == [(x.query_name, x.ref_id, x.pos, x.flag) for x in rb]
    assert not [f for f in os.listdir(tmp_path) if "sorttmp" in f]
    assert open(paths[0] + ".bai", "rb").read()[:4] == b"BAI\x01" and os.path.getsize(paths[1] + ".bai") > 0


def test_index_of_unaligned_bam_and_errors(tmp_path):
    path = str(tmp_path / "u.bam")
    with bamio.BamWriter(path, "@HD\tVN:1.5\tSO:unknown\n", []) as w:
        for i in range(7):
            w.write(bamio.BamRecord("m/%d/ccs" % i, flag=4, seq="ACGT" * 10))
    assert bamnative.sort_and_index(path) is False                  # unaligned reads are already "sorted": input order is kept
    assert open(path + ".bai", "rb").read() == b"BAI\x01" + (0).to_bytes(4, "little") + (7).to_bytes(8, "little")
    with pytest.raises(IOError):
        bamnative.index_build(str(tmp_path / "missing.bam"))
    # a memory limit below a single record: every record becomes its own run, the merge still sorts
    p2 = str(tmp_path / "b.bam")
    with bamio.BamWriter(p2, "", [("c", 1000)]) as w:
        w.write(bamio.BamRecord("a", flag=0, ref_id=0, pos=500, cigar=[(0, 4)], seq="ACGT"))
        w.write(bamio.BamRecord("b", flag=0, ref_id=0, pos=100, cigar=[(0, 4)], seq="ACGT"))
    bamnative._check(bamnative.load().ccsm_bam_sort(p2.encode(), (p2 + ".s").encode(), 1, 6, 10))
    with bamio.BamReader(p2 + ".s") as rd:
        assert [r.query_name for r in rd] == ["b", "a"]
    with pytest.raises(IOError):                                        # an output path that cannot be created
        bamnative._check(bamnative.load().ccsm_bam_sort(p2.encode(), str(tmp_path / "no_dir" / "x.bam").encode(), 1, 6, 0))


def test_modcalls_index_sort_survive_corrupted_records(tmp_path):
    """The same byte-corruption fuzz over the newer parsers: ccsm_bam_modcalls_of_batch (MM/ML, CIGAR walk, with and without
    reference site masks), ccsm_bam_align_info, ccsm_bam_index_build and ccsm_bam_sort either work or raise IOError."""
    import subprocess
    import sys
    from conftest import GOLDEN, ROOT
    script = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from ccsmeth_amd import bamio, bamnative
rng = np.random.default_rng(9)
src, mut, outp = %r, %r, %r
raw = b"".join(bamio.bgzf_blocks(open(src, "rb")))
masks = [np.ones(6000, np.uint8) * 3, np.ones(5200, np.uint8) * 3, None]
ok = err = rows = 0
for trial in range(120):
    b = bytearray(raw)
    lo = 200 if trial %% 3 else 0                      # most trials leave the header intact so that records are reached
    for _ in range(int(rng.integers(1, 8))):
        b[int(rng.integers(lo, len(b)))] = int(rng.integers(0, 256))
    with open(mut, "wb") as f:
        for i in range(0, len(b), 0xff00):
            f.write(bamio.bgzf_compress_block(bytes(b[i:i + 0xff00]), 1))
        f.write(bamio._BGZF_EOF)
    try:
        with bamnative.NativeBamReader(mut, threads=2) as rd:
            while True:
                bt = rd.next_batch(16)
                if bt is None:
                    break
                for kw in (dict(), dict(refsites_all=True, site_masks=masks[:rd.n_ref] + [None] * max(0, rd.n_ref - 3), base_clip=3)):
                    out = bamnative.modcalls_of_batch(bt, threads=2, **kw)
                    rows += len(out[0])
                    assert all(len(a) == len(out[0]) for a in out[:5])
                bamnative.align_info(bt)
                bt.close()
        bamnative.index_build(mut, mut + ".bai", threads=2)
        bamnative._check(bamnative.load().ccsm_bam_sort(mut.encode(), outp.encode(), 2, 1, 1 << 28))
        ok += 1
    except IOError:
        err += 1
print("ok", ok, "ioerror", err, "rows", rows)
assert ok + err == 120 and ok > 0 and rows > 10000
''' % (ROOT, os.path.join(GOLDEN, "freqb", "aligned.modbam.bam"), str(tmp_path / "mut.bam"), str(tmp_path / "out.bam"))
    res = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
