"""Scan-free chunked reading (ccsm_bam_seek_chunk) and the writer-side index (ccsm_bam_writer_track_index / take_index /
ccsm_bam_index_write) of libccsm_bam: what the multi-GPU call_mods is built on (reference: one reader process fills a
shared queue, extract_features.py:129-177, and the modbam is indexed by re-reading it, call_modifications.py:592-607).  CPU only."""
import os

import numpy as np
import pytest

from ccsmeth_amd import bamio, bamnative
from test_bamnative import _aligned_records, _bai_parse, _make_bam


def _sequential(path):
    """[(virtual offset, raw record bytes, name hash)] of every record + the offset behind the last, read in one pass"""
    out = []
    with bamnative.NativeBamReader(path, threads=2) as rd:
        first = rd.tell()
        while True:
            v = rd.tell()
            b = rd.next_batch(1)
            if b is None:
                break
            out.append((v, bytes(b.records), int(b.name_hash[0])))
            b.close()
    return first, out


@pytest.mark.parametrize("chunk", [700, 4096, 65536, 300000, 1 << 30])
@pytest.mark.parametrize("level", [1, 6])
def test_chunks_partition_the_file_exactly(tmp_path, chunk, level):
    """Every record belongs to exactly one chunk, chunk after chunk gives the file's order, and the hand-over offsets chain:
    where chunk k's last record ends is where the next non-empty chunk was found to begin.  Chunk sizes from far below a BGZF block
    (most chunks empty, boundaries inside blocks, records spanning many chunks) to larger than the file."""
    rng = np.random.default_rng(5)
    path = str(tmp_path / "in.bam")
    recs = []
    with bamio.BamWriter(path, "@HD\tVN:1.5\tSO:unknown\n", [("chr1", 100000)], level=level) as w:
        for i in range(60):
            L = int(rng.choice([0, 5, 300, 2000, 30000, 90000]))
            seq = "".join(rng.choice(list("ACGT"), size=L))
            kin = lambda: rng.integers(0, 256, L).astype(np.uint8)  # noqa: E731
            # kinetics full of 0xff / 0x00 runs and of bytes that look like record fields
            fi = kin()
            if L > 100:
                fi[10:60] = 0xff
                fi[60:96] = np.frombuffer(np.array([40, -1, -1, 0x4802, 0, 4, 0, -1, -1], "<i4").tobytes(), np.uint8)
            r = bamio.BamRecord("m/%d/ccs" % i, flag=4, seq=seq, qual=rng.integers(0, 60, L).astype(np.uint8),
                                tags=[("fi", "BC", fi), ("fp", "BC", kin()), ("ri", "BC", kin()), ("rp", "BC", kin()), ("fn", "C", 9), ("rn", "C", 8)])
            recs.append(r)
            w.write(r)
    first, seq_recs = _sequential(path)
    size = os.path.getsize(path)
    got, chain = [], []
    with bamnative.NativeBamReader(path, threads=3) as rd:
        for k in range((size + chunk - 1) // chunk + 1):          # one chunk past the end: must be empty
            v0 = rd.seek_chunk(k * chunk, (k + 1) * chunk)
            if v0 == 0:
                continue
            n = 0
            while True:
                v = rd.tell()
                b = rd.next_batch(int(rng.integers(1, 9)))
                if b is None:
                    break
                for j in range(b.n_reads):
                    got.append((bytes(b.records[b.rec_offset[j]:b.rec_offset[j + 1]]), int(b.name_hash[j])))
                assert (b.voffset_start >> 16) >= k * chunk and (b.voffset_start >> 16) < (k + 1) * chunk and b.voffset_start == v
                n += b.n_reads
                b.close()
            assert n > 0
            chain.append((v0, rd.tell()))
    assert [g[0] for g in got] == [s[1] for s in seq_recs]
    assert [g[1] for g in got] == [s[2] for s in seq_recs]
    assert chain[0][0] == first
    for (a0, a1), (b0, b1) in zip(chain, chain[1:]):
        assert a1 == b0
    assert len(set(s[2] for s in seq_recs)) == len(seq_recs)      # the name hashes are distinct keys


def test_chunks_of_the_mixed_fixture(tmp_path):
    """The reader fixture of test_bamnative (aligned and unaligned records, every tag type, broken kinetics) in chunks."""
    rng = np.random.default_rng(11)
    path = str(tmp_path / "in.bam")
    _make_bam(path, rng, n_reads=120)
    _, seq_recs = _sequential(path)
    for chunk in (1500, 50000):
        got = []
        with bamnative.NativeBamReader(path, threads=2) as rd:
            for k in range(os.path.getsize(path) // chunk + 1):
                if rd.seek_chunk(k * chunk, (k + 1) * chunk) == 0:
                    continue
                while True:
                    b = rd.next_batch(7)
                    if b is None:
                        break
                    got += [bytes(b.records[b.rec_offset[j]:b.rec_offset[j + 1]]) for j in range(b.n_reads)]
                    b.close()
        assert got == [s[1] for s in seq_recs]


def test_zlib_and_libdeflate_files_are_interchangeable(tmp_path, monkeypatch):
    """The native writer's blocks (libdeflate when the runtime library is there) inflate with zlib (bamio) and the other way round."""
    import subprocess
    import sys
    rng = np.random.default_rng(2)
    src = str(tmp_path / "py.bam")
    _make_bam(src, rng, n_reads=30)
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from ccsmeth_amd import bamnative as bn\n"
            "rd = bn.NativeBamReader(%r, threads=2); w = bn.NativeBamWriter(sys.argv[1], rd.header_text, rd.raw_refs, rd.n_ref, threads=2)\n"
            "b = rd.next_batch(1000); w.write_batch(b); w.close()\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), src)
    outs = {}
    for name, env in (("deflate", {}), ("zlib", {"CCSM_BAM_ZLIB": "1"})):
        out = str(tmp_path / (name + ".bam"))
        subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, **env))
        with bamio.BamReader(out) as rd:
            outs[name] = [(r.query_name, r.flag, r.seq, [(t[0], t[1]) for t in r.tags]) for r in rd]
    assert outs["deflate"] == outs["zlib"] and len(outs["zlib"]) == 30


def _write_runs(path, recs_by_run, refs, track=True):
    """Write records run by run through the native writer (flush between runs); -> (header_end, [IndexRun per run])"""
    tmp = path + ".src"
    text = "@HD\tVN:1.5\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    runs = []
    with bamio.BamWriter(tmp, text, refs) as w:
        for rr in recs_by_run:
            for r in rr:
                w.write(r)
    with bamnative.NativeBamReader(tmp, threads=2) as rd, bamnative.NativeBamWriter(path, rd.header_text, rd.raw_refs, rd.n_ref, threads=2) as wr:
        header_end = wr.flush()
        wr.track_index(track)
        for rr in recs_by_run:
            if rr:
                b = rd.next_batch(len(rr))
                wr.write_batch(b, rm_pulse=False)
                b.close()
            wr.flush()
            if track:
                runs.append(wr.take_index())
    os.remove(tmp)
    return header_end, runs


def test_writer_side_index_equals_the_streamed_one(tmp_path):
    """Sorted aligned records (+ a placed-unmapped and unplaced ones) written in 9 runs: the index built from the writer's run tables
    is byte for byte the one ccsm_bam_index_build makes by re-reading the file."""
    rng = np.random.default_rng(3)
    recs = _aligned_records(rng, 500)
    key = lambda r: ((r.ref_id if r.ref_id >= 0 else 1 << 31), r.pos + 1, 1 if r.flag & 16 else 0)  # noqa: E731
    recs.sort(key=key)
    refs = [("c%d" % i, 300000) for i in range(3)]
    cuts = sorted(rng.choice(np.arange(1, len(recs)), 8, replace=False).tolist())
    by_run = [recs[a:b] for a, b in zip([0] + cuts, cuts + [len(recs)])]
    path = str(tmp_path / "w.bam")
    _, runs = _write_runs(path, by_run, refs)
    assert sum(r.n_records for r in runs) == len(recs) and all(r.sorted for r in runs)
    ok, n = bamnative.index_write(path + ".w.bai", 3, runs)
    assert ok and n == len(recs)
    ok2, n2 = bamnative.index_build(path, path + ".s.bai", threads=2)
    assert ok2 and n2 == len(recs)
    a, b = _bai_parse(path + ".w.bai"), _bai_parse(path + ".s.bai")
    assert a[1] == b[1] == 5
    for (bins_w, lin_w), (bins_s, lin_s) in zip(a[0], b[0]):
        assert lin_w == lin_s and set(bins_w) == set(bins_s)
        for k in bins_w:
            if k == 37450:
                assert bins_w[k][1] == bins_s[k][1] and bins_w[k][0][0] == bins_s[k][0][0]
            else:
                assert [c[0] for c in bins_w[k]] == [c[0] for c in bins_s[k]]       # chunk starts are record starts: identical
                for cw, cs in zip(bins_w[k], bins_s[k]):                            # chunk ends name the same file position
                    assert cw[1] == cs[1] or ((cw[1] & 0xffff) and (cs[1] & 0xffff) == 0 and cs[1] > cw[1])
    # an unsorted sequence of runs is reported, and nothing is written
    ok3, _ = bamnative.index_write(path + ".x.bai", 3, runs[::-1])
    assert not ok3 and not os.path.exists(path + ".x.bai")


def test_index_of_stitched_runs_with_shifts(tmp_path):
    """Two writers ("ranks") hold alternating runs of a sorted file; the runs are stitched into input order and the index is built
    from the run tables with each run's shift: equal to indexing the stitched file from scratch."""
    rng = np.random.default_rng(8)
    recs = _aligned_records(rng, 300)
    key = lambda r: ((r.ref_id if r.ref_id >= 0 else 1 << 31), r.pos + 1, 1 if r.flag & 16 else 0)  # noqa: E731
    recs.sort(key=key)
    refs = [("c%d" % i, 300000) for i in range(3)]
    cuts = list(range(40, len(recs), 40))
    by_run = [recs[a:b] for a, b in zip([0] + cuts, cuts + [len(recs)])]
    parts, tables, hdr_end = [], [], None
    for rank in range(2):
        p = str(tmp_path / ("part%d.bam" % rank))
        he, runs = _write_runs(p, [rr for i, rr in enumerate(by_run) if i % 2 == rank], refs)
        parts.append(p)
        tables.append(runs)
        hdr_end = he
    order = [(i % 2, i // 2) for i in range(len(by_run))]
    out = str(tmp_path / "stitched.bam")
    spans = [(parts[rk], tables[rk][j].file_start, tables[rk][j].file_end) for rk, j in order]
    dst = bamnative.stitch_runs_parallel(out, parts[0], hdr_end, spans, mine=range(len(spans)), finish=True)
    shifts = [d - tables[rk][j].file_start for d, (rk, j) in zip(dst, order)]
    ok, n = bamnative.index_write(out + ".bai", 3, [tables[rk][j] for rk, j in order], shifts)
    assert ok and n == len(recs)
    ok2, _ = bamnative.index_build(out, out + ".ref.bai", threads=2)
    assert ok2
    a, b = _bai_parse(out + ".bai"), _bai_parse(out + ".ref.bai")
    assert a[1] == b[1]
    for (bins_w, lin_w), (bins_s, lin_s) in zip(a[0], b[0]):
        assert lin_w == lin_s and set(bins_w) == set(bins_s)
        for k in bins_w:
            assert [c[0] for c in bins_w[k]] == [c[0] for c in bins_s[k]] or k == 37450
    with bamio.BamReader(out) as rd:
        assert [r.query_name for r in rd] == [r.query_name for r in recs]


def _rewrap_truncated(src, dst, cut_bytes):
    """`src` with the last `cut_bytes` of its INFLATED payload removed, as well-formed BGZF blocks + the EOF marker: a BAM whose last
    record is cut in the middle while every block is intact (a damaged upload re-compressed, a writer killed between records' halves)."""
    import gzip
    import struct
    import zlib
    payload = gzip.open(src, "rb").read()
    payload = payload[:len(payload) - cut_bytes]
    with open(dst, "wb") as f:
        for a in list(range(0, len(payload), 60000)) + [None]:
            chunk = b"" if a is None else payload[a:a + 60000]
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            cd = co.compress(chunk) + co.flush()
            f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(cd) + 25) + cd
                    + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))


@pytest.mark.parametrize("chunk", [3000, 200000, 1 << 30])
def test_hand_over_chain_ends_at_the_end_of_the_data(tmp_path, chunk):
    """ccsm_bam_eof_voffset = what tell() reports behind the last record (from the block headers alone), and sharding.verify_chain
    closes the chain on it: on an intact file the chunked path's chain ends exactly there; on a file whose last record is cut in the
    middle (every BGZF block intact) the sequential reader raises and so does the chain - the chunk the cut record starts in is either
    reported empty by the record search or raises itself; before this check it was silently dropped (sharding.py)."""
    from ccsmeth_amd import sharding
    rng = np.random.default_rng(23)
    path = str(tmp_path / "in.bam")
    _make_bam(path, rng, n_reads=40)
    first, seq_recs = _sequential(path)

    def chunked(p):
        log = []
        with bamnative.NativeBamReader(p, threads=2) as rd:
            eof = rd.eof_voffset()
            nch = sharding.n_chunks_of(os.path.getsize(p), chunk)
            for k in range(nch):
                v0 = rd.seek_chunk(k * chunk, (k + 1) * chunk)
                if v0 == 0:
                    log.append((k, 0, 0, 0))
                    continue
                n = 0
                while True:
                    b = rd.next_batch(5)
                    if b is None:
                        break
                    n += b.n_reads
                    b.close()
                log.append((k, v0, rd.tell(), n))
        return eof, nch, log

    eof, nch, log = chunked(path)
    with bamnative.NativeBamReader(path, threads=1) as rd:
        while True:
            b = rd.next_batch(64)
            if b is None:
                break
            b.close()
        assert rd.tell() == eof                                  # the definition: a reader behind the last record stands at eof_voffset
    assert sharding.verify_chain(first, log, n_chunks=nch, eof_voffset=eof) == len(seq_recs)
    with pytest.raises(RuntimeError, match="never reported"):
        sharding.verify_chain(first, log[:-1], n_chunks=nch, eof_voffset=eof)
    # the same file with its last record cut in the middle
    cut = str(tmp_path / "cut.bam")
    _rewrap_truncated(path, cut, len(seq_recs[-1][1]) // 2)
    with pytest.raises(IOError):                    # the sequential reader: an error, not a short file
        _sequential(cut)
    first_c = None
    with bamnative.NativeBamReader(cut, threads=1) as rd:
        first_c = rd.tell()
    try:
        eof_c, nch_c, log_c = chunked(cut)
    except IOError:
        return                                                   # the chunk holding the cut record raised itself: also not silent
    with pytest.raises(RuntimeError, match="truncated BAM record"):
        sharding.verify_chain(first_c, log_c, n_chunks=nch_c, eof_voffset=eof_c)
