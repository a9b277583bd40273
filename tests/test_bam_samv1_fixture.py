"""libccsm_bam against a BAM laid out BY HAND from the SAM/BAM format specification (SAMv1 sections 4.1 BGZF, 4.2 BAM, 4.2.4 tag
value types, 5.1.1 / 5.2 BAI, 5.3 reg2bin), not by ccsmeth_amd/bamio.py: nothing in this file imports bamio.  pysam/htslib is not
in the image and cannot be installed (no network), so the interoperability anchor is the published byte layout itself:
  * SPEC_R001 is record r001 of the specification's example alignment (section 1.1) written out as literal bytes, field by field
    from the table of section 4.2; the small encoder below must reproduce it before it is trusted for the other records;
  * the 28-byte BGZF end-of-file block is the constant of section 4.1.2;
  * the bin numbers are the closed forms of section 5.3 (bin 4681 + (pos >> 14) for an interval inside one 16 kb window, 4680 for an
    unplaced read, 37450 = the samtools metadata pseudo-bin).
The reader must hand back these records byte for byte, the modbam writer's output must parse with the spec-derived decoder below
(MM as Z, ML as B:C, refilled tags with their original type codes), and sort + index must give the spec's bins and linear index."""
import struct
import zlib

import numpy as np
import pytest

from ccsmeth_amd import bamnative

SEQ_CODES = "=ACMGRSVTWYHKDBN"          # 4.2: 4-bit base encoding
CIGAR_OPS = "MIDNSHP=X"                 # 4.2: op in the low 4 bits, length in the upper 28
EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")     # 4.1.2

# r001  99  ref  7  30  8M2I4M1D3M  =  37  39  TTAGATAAAGGATACTG  *        (section 1.1; 1-based POS 7 -> 0-based 6)
SPEC_R001 = bytes.fromhex(
    "53000000"                  # block_size = 32 + 5 + 5*4 + 9 + 17 = 83
    "00000000" "06000000"       # refID 0, pos 6
    "05" "1e" "4912"            # l_read_name 5, mapq 30, bin 4681 = 0x1249
    "0500" "6300"               # n_cigar_op 5, flag 99
    "11000000"                  # l_seq 17
    "00000000" "24000000" "27000000"    # next_refID 0 ('='), next_pos 36, tlen 39
    "7230303100"                # "r001\0"
    "80000000" "21000000" "40000000" "12000000" "30000000"      # 8M 2I 4M 1D 3M
    "88" "14" "18" "11" "14" "41" "81" "28" "40"                # TT AG AT AA AG GA TA CT G-
    + "ff" * 17)                # QUAL '*'


# ---- a spec-derived encoder / decoder (test-local, independent of the product code) ----------------------------------------------
def bin_of(beg, end):
    """Section 5.3, reg2bin (0-based half-open interval)."""
    end -= 1
    for shift, first in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return first + (beg >> shift)
    return 0


def tag_bytes(tag, typ, val):
    out = tag.encode()
    if typ in "cCsSiI":
        return out + typ.encode() + struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I"}[typ], val)
    if typ == "f":
        return out + b"f" + struct.pack("<f", val)
    if typ == "A":
        return out + b"A" + val.encode()
    if typ in "ZH":
        return out + typ.encode() + val.encode() + b"\0"
    if typ[0] == "B":
        sub = typ[1]
        fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]
        return out + b"B" + sub.encode() + struct.pack("<i", len(val)) + struct.pack("<%d%s" % (len(val), fmt), *val)
    raise ValueError(typ)


def record_bytes(name, flag, tid, pos, mapq, cigar, seq, qual=None, tags=(), mate=(-1, -1, 0)):
    ops = [(int(n), CIGAR_OPS.index(o)) for n, o in cigar]
    reflen = sum(n for n, o in ops if o in (0, 2, 3, 7, 8))
    b = bin_of(pos, pos + reflen) if pos >= 0 and reflen else bin_of(pos, pos + 1) if pos >= 0 else 4680
    codes = [SEQ_CODES.index(c) for c in seq] + [0]
    packed = bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(seq), 2))
    q = bytes([0xFF] * len(seq)) if qual is None else bytes(qual)
    body = struct.pack("<iiBBHHHiiii", tid, pos, len(name) + 1, mapq, b, len(ops), flag, len(seq), *mate)
    body += name.encode() + b"\0" + b"".join(struct.pack("<I", (n << 4) | o) for n, o in ops) + packed + q
    body += b"".join(tag_bytes(*t) for t in tags)
    return struct.pack("<i", len(body)) + body


def bgzf_block(payload, level=6):
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = c.compress(payload) + c.flush()
    head = bytes.fromhex("1f8b08040000000000ff0600424302 00".replace(" ", "")) + struct.pack("<H", len(comp) + 25)
    return head + comp + struct.pack("<II", zlib.crc32(payload), len(payload))


def bam_file(path, text, refs, records, per_block=3):
    head = b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs))
    for nm, ln in refs:
        head += struct.pack("<i", len(nm) + 1) + nm.encode() + b"\0" + struct.pack("<i", ln)
    with open(path, "wb") as fh:
        fh.write(bgzf_block(head))
        for i in range(0, len(records), per_block):      # records never straddle blocks here; the reader test elsewhere covers that
            fh.write(bgzf_block(b"".join(records[i:i + per_block])))
        fh.write(EOF_BLOCK)


def inflate_all(path):
    raw, out, off, blocks = open(path, "rb").read(), bytearray(), 0, []
    while off < len(raw):
        assert raw[off:off + 4] == b"\x1f\x8b\x08\x04" and raw[off + 12:off + 16] == b"BC\x02\x00"
        bsize = struct.unpack_from("<H", raw, off + 16)[0] + 1
        payload = zlib.decompress(raw[off + 18:off + bsize - 8], -15)
        crc, isize = struct.unpack_from("<II", raw, off + bsize - 8)
        assert crc == zlib.crc32(payload) and isize == len(payload)
        blocks.append((off, len(out), len(payload)))
        out += payload
        off += bsize
    assert raw.endswith(EOF_BLOCK)
    return bytes(out), blocks


def parse_tags(buf):
    out, off = [], 0
    size = {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4, "A": 1}
    fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}
    while off < len(buf):
        tag, typ = buf[off:off + 2].decode(), chr(buf[off + 2])
        off += 3
        if typ == "A":
            out.append((tag, "A", chr(buf[off]))); off += 1
        elif typ in fmt:
            out.append((tag, typ, struct.unpack_from("<" + fmt[typ], buf, off)[0])); off += size[typ]
        elif typ in "ZH":
            end = buf.index(b"\0", off)
            out.append((tag, typ, buf[off:end].decode())); off = end + 1
        elif typ == "B":
            sub = chr(buf[off]); n = struct.unpack_from("<i", buf, off + 1)[0]
            out.append((tag, "B" + sub, list(struct.unpack_from("<%d%s" % (n, fmt[sub]), buf, off + 5)))); off += 5 + n * size[sub]
        else:
            raise AssertionError("tag type %r is not in SAMv1 4.2.4" % typ)
    return out


def parse_bam(path):
    data, blocks = inflate_all(path)
    assert data[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", data, 4)[0]
    text = data[8:8 + l_text].decode()
    off = 8 + l_text
    n_ref = struct.unpack_from("<i", data, off)[0]; off += 4
    refs = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", data, off)[0]
        refs.append((data[off + 4:off + 4 + ln - 1].decode(), struct.unpack_from("<i", data, off + 4 + ln)[0]))
        off += 8 + ln
    recs = []
    while off < len(data):
        bs = struct.unpack_from("<i", data, off)[0]
        tid, pos, l_name, mapq, bin_, n_cig, flag, l_seq, ntid, npos, tlen = struct.unpack_from("<iiBBHHHiiii", data, off + 4)
        p = off + 36
        name = data[p:p + l_name - 1].decode(); p += l_name
        cigar = [(c >> 4, CIGAR_OPS[c & 15]) for c in struct.unpack_from("<%dI" % n_cig, data, p)]; p += 4 * n_cig
        nib = data[p:p + (l_seq + 1) // 2]; p += (l_seq + 1) // 2
        seq = "".join(SEQ_CODES[(nib[i >> 1] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq))
        qual = data[p:p + l_seq]; p += l_seq
        recs.append(dict(name=name, flag=flag, tid=tid, pos=pos, mapq=mapq, bin=bin_, cigar=cigar, seq=seq, qual=qual,
                         tags=parse_tags(data[p:off + 4 + bs]), raw=data[off:off + 4 + bs], data_off=off))
        off += 4 + bs
    return text, refs, recs, blocks


COMP = str.maketrans("ACGTN", "TGCAN")


def spec_records():
    """The six records of the specification's example + HiFi reads carrying the kinetics tags the hot path consumes
    (extract_features.py:98-127 reads fi/ri/fp/rp as B:C arrays, fn/rn as integers of whatever width the writer chose)."""
    ex = [
        record_bytes("r001", 99, 0, 6, 30, [(8, "M"), (2, "I"), (4, "M"), (1, "D"), (3, "M")], "TTAGATAAAGGATACTG", mate=(0, 36, 39)),
        record_bytes("r002", 0, 0, 8, 30, [(3, "S"), (6, "M"), (1, "P"), (1, "I"), (4, "M")], "AAAAGATAAGGATA"),
        record_bytes("r003", 0, 0, 8, 30, [(5, "S"), (6, "M")], "GCCTAAGCTAA", tags=[("SA", "Z", "ref,29,-,6H5M,17,0;")]),
        record_bytes("r004", 0, 0, 15, 30, [(6, "M"), (14, "N"), (5, "M")], "ATAGCTTCAGC"),
        record_bytes("r003", 2064, 0, 28, 17, [(6, "H"), (5, "M")], "TAGGC", tags=[("SA", "Z", "ref,9,+,5S6M,30,1;")]),
        record_bytes("r001", 147, 0, 36, 30, [(9, "M")], "CAGCGGCAT", tags=[("NM", "i", 1)], mate=(0, 6, -39)),
    ]
    rng = np.random.default_rng(41)
    hifi, meta = [], []
    for i, (L, flag, fn_t, rn_t) in enumerate([(45, 4, "C", "C"), (64, 4, "S", "c"), (33, 16, "i", "s"), (51, 0, "I", "C")]):
        seq = "".join(rng.choice(list("ACGT"), L - 12)) + "ACGTCGCGAACG"          # odd and even lengths, CpGs near the end
        seq = "".join(rng.permutation(list(seq[:L - 12]))) + seq[L - 12:]
        kin = {t: rng.integers(0, 256, L).tolist() for t in ("fi", "ri", "fp", "rp")}
        fn, rn = int(rng.integers(5, 90)), int(rng.integers(5, 90))
        tags = [("zm", "i", 4711 + i), ("np", "C" if i % 2 else "S", 11), ("rq", "f", 0.9995), ("sn", "Bf", [9.5, 10.25, 3.0, 7.125]),
                ("fi", "BC", kin["fi"]), ("fn", fn_t, fn), ("fp", "BC", kin["fp"]), ("ri", "BC", kin["ri"]), ("rn", rn_t, rn),
                ("rp", "BC", kin["rp"]), ("RG", "Z", "hifi"), ("XA", "A", "q"), ("XH", "H", "1AE301"), ("Xs", "Bs", [-3, 200, 7])]
        aligned = not flag & 4
        hifi.append(record_bytes("m64011_190830_220126/%d/ccs" % (101 + i), flag, 0 if aligned else -1, 2 + i if aligned else -1,
                                 60 if aligned else 255, [(L, "M")] if aligned else [], seq,
                                 qual=rng.integers(0, 94, L).tolist(), tags=tags))
        meta.append(dict(seq=seq, flag=flag, fn=fn, rn=rn, tags=tags, **kin))
    return ex, hifi, meta


def test_encoder_reproduces_the_specification_example_bytes():
    ex, _, _ = spec_records()
    assert ex[0] == SPEC_R001
    assert len(EOF_BLOCK) == 28 and zlib.decompress(EOF_BLOCK[18:-8], -15) == b""
    # section 5.3 worked values: one 16 kb window, a 128 kb window, the whole 512 Mb range, the unplaced-read convention
    assert bin_of(6, 22) == 4681 and bin_of(16384, 16385) == 4682 and bin_of(16383, 16385) == 585
    assert bin_of(0, 1 << 29) == 0 and bin_of((1 << 26) - 1, (1 << 26) + 1) == 0 and bin_of(1 << 26, (1 << 26) + 1) == 4681 + 4096
    assert bin_of(-1, 0) == 4680


def test_reader_returns_hand_laid_out_records_byte_for_byte(tmp_path):
    ex, hifi, meta = spec_records()
    path = str(tmp_path / "spec.bam")
    text = "@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:ref\tLN:45\n@RG\tID:hifi\tPL:PACBIO\n"
    bam_file(path, text, [("ref", 45)], ex + hifi)
    with bamnative.NativeBamReader(path, threads=2) as rd:
        assert rd.header_text == text and rd.n_ref == 1
        assert rd.raw_refs == bytes.fromhex("04000000") + b"ref\0" + bytes.fromhex("2d000000")
        b = rd.next_batch(100)
        assert rd.next_batch(100) is None
    assert b.n_reads == 10
    raws = [bytes(b.records[b.rec_offset[k]:b.rec_offset[k + 1]]) for k in range(10)]
    assert raws == ex + hifi and raws[0] == SPEC_R001
    assert list(b.flag) == [99, 0, 0, 0, 2064, 147, 4, 4, 16, 0]
    assert list(b.length[:6]) == [0] * 6                     # no kinetics: not usable by the hot path
    for k, m in enumerate(meta, start=6):
        L, o = len(m["seq"]), int(b.offset[k])
        assert b.length[k] == L
        # the batch is in the ORIENTATION OF THE READ (get_forward_sequence / the reference reverses fi.. of reverse-strand alignments
        # is NOT done: extract_features.py:98-127 takes the tag arrays as stored)
        fwd = m["seq"] if not m["flag"] & 16 else m["seq"].translate(COMP)[::-1]
        assert bytes(b.seq[o:o + L]).decode() == fwd
        for t in ("fi", "ri", "fp", "rp"):
            assert list(b.fi if t == "fi" else b.ri if t == "ri" else b.fp if t == "fp" else b.rp)[o:o + L] == m[t]
        assert (b.fn[k], b.rn[k]) == (m["fn"], m["rn"])
    b.close()


def test_modbam_writer_output_parses_with_the_spec_decoder(tmp_path):
    ex, hifi, meta = spec_records()
    inp, outp = str(tmp_path / "in.bam"), str(tmp_path / "out.bam")
    text = "@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:ref\tLN:45\n"
    bam_file(inp, text, [("ref", 45)], ex[:2] + hifi)
    with bamnative.NativeBamReader(inp, threads=1) as rd, \
            bamnative.NativeBamWriter(outp, rd.header_text, rd.raw_refs, rd.n_ref, threads=2, level=5) as wr:
        b = rd.next_batch(100)
        first, locs, probs, tagged = [0], [], [], []
        want = []
        for k in range(b.n_reads):
            if k < 2:
                tagged.append(0); first.append(first[-1]); want.append(None)
                continue
            m = meta[k - 2]
            fwd = m["seq"] if not m["flag"] & 16 else m["seq"].translate(COMP)[::-1]
            cpg = [j for j in range(len(fwd) - 1) if fwd[j:j + 2] == "CG"]
            pr = [0.0, 0.5, 0.999, 1.0, 0.25][:len(cpg)] + [0.7] * max(0, len(cpg) - 5)
            locs += cpg; probs += pr; tagged.append(1); first.append(first[-1] + len(cpg))
            # MM: number of unmodified C's skipped before each called one (SAM tags specification, "Base modifications")
            cs = [j for j, c in enumerate(fwd) if c == "C"]
            deltas, prev = [], -1
            for j in cpg:
                deltas.append(cs.index(j) - prev - 1); prev = cs.index(j)
            want.append(("C+m?," + ",".join(map(str, deltas)) + ";", [min(int(p * 256), 255) for p in pr]))
        n = wr.write_batch(b, np.array(first, np.int32), np.array(locs, np.int32), np.array(probs, np.float32),
                           np.array(tagged, np.uint8), rm_pulse=True)
        b.close()
    assert n == 4
    otext, orefs, recs, _ = parse_bam(outp)
    assert otext == text and orefs == [("ref", 45)] and len(recs) == 6
    assert recs[0]["raw"] == SPEC_R001 and recs[1]["raw"] == ex[1]                         # untouched records pass through unchanged
    itext, _, irecs, _ = parse_bam(inp)
    for k in range(2, 6):
        o, i, (mm, ml) = recs[k], irecs[k], want[k]
        assert [o[f] for f in ("name", "flag", "tid", "pos", "mapq", "bin", "cigar", "seq", "qual")] == \
               [i[f] for f in ("name", "flag", "tid", "pos", "mapq", "bin", "cigar", "seq", "qual")]
        kept = [t for t in i["tags"] if t[0] not in ("fi", "fp", "ri", "rp", "MM", "ML")]    # rm_pulse (_bam2modbam.py refill)
        assert o["tags"][:len(kept)] == kept                                                # same order, same TYPE CODES, same values
        assert [t[:2] for t in o["tags"][len(kept):]] == [("MM", "Z"), ("ML", "BC")]
        got_mm, got_ml = o["tags"][-2][2], o["tags"][-1][2]
        assert got_mm.split(",")[1:] == mm.split(",")[1:] and got_mm.split(",")[0] in ("C+m", "C+m?", "C+m.")
        assert got_ml == ml
        assert struct.unpack_from("<i", o["raw"], 0)[0] == len(o["raw"]) - 4


def test_sort_and_index_give_the_specification_bins(tmp_path):
    ex, hifi, _ = spec_records()
    # three references so that the linear index and the per-reference sections are exercised; positions chosen around bin borders
    refs = [("ref", 45), ("big", 1 << 28), ("mid", 500000)]
    recs = list(ex)
    spots = [(1, 0, 100), (1, 16383, 2), (1, 16384, 50), (1, (1 << 26) - 10, 20), (1, 1 << 26, 1000), (1, 131071, 3), (2, 70000, 70000),
             (2, 16000, 800), (2, 499000, 900)]
    for n, (tid, pos, ln) in enumerate(spots):
        cigar, seq = ([(ln, "M")], "ACGT" * 8) if ln <= 30 else ([(10, "M"), (ln - 20, "N"), (10, "M")], "ACGTACGTAC" * 2)
        recs.append(record_bytes("s%d" % n, 16 if n % 2 else 0, tid, pos, 40, cigar, seq[:sum(k for k, o in cigar if o == "M")],
                                 tags=[("NM", "i", n)]))
    recs.append(hifi[0]); recs.append(hifi[1])                                              # unplaced reads go last
    order = np.random.default_rng(5).permutation(len(recs))
    path = str(tmp_path / "x.bam")
    bam_file(path, "@HD\tVN:1.6\tSO:unsorted\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs), refs, [recs[i] for i in order], per_block=2)
    assert bamnative.sort_and_index(path, threads=2) is True
    text, orefs, out, blocks = parse_bam(path)
    assert text.startswith("@HD\tVN:1.6\tSO:coordinate\n") and orefs == refs and len(out) == len(recs)
    keys = [((r["tid"] if r["tid"] >= 0 else 1 << 31), r["pos"]) for r in out]
    assert keys == sorted(keys)
    assert sorted(r["raw"] for r in out) == sorted(recs)                                    # every record survives bit for bit

    def voff(data_off):
        for c, a, n in blocks:
            if a <= data_off < a + n:
                return (c << 16) | (data_off - a)
        raise AssertionError

    bai = open(path + ".bai", "rb").read()
    assert bai[:4] == b"BAI\x01" and struct.unpack_from("<i", bai, 4)[0] == 3
    off = 8
    for tid in range(3):
        mine = [r for r in out if r["tid"] == tid]
        want_bins = {}
        for r in mine:
            reflen = sum(n for n, o in r["cigar"] if o in "MDN=X")
            assert r["bin"] == bin_of(r["pos"], r["pos"] + reflen)                          # the records' own bin field (4.2)
            want_bins.setdefault(r["bin"], []).append(voff(r["data_off"]))
        n_bin = struct.unpack_from("<i", bai, off)[0]; off += 4
        got_bins = {}
        for _ in range(n_bin):
            bn, n_chunk = struct.unpack_from("<Ii", bai, off); off += 8
            got_bins[bn] = [struct.unpack_from("<QQ", bai, off + 16 * k) for k in range(n_chunk)]
            off += 16 * n_chunk
        meta = got_bins.pop(37450)                                                          # 5.2: pseudo-bin, (ref_beg, ref_end), (n_mapped, n_unmapped)
        assert meta[1] == (len(mine), 0) and meta[0][0] == voff(mine[0]["data_off"])
        assert set(got_bins) == set(want_bins)
        for bn, vs in want_bins.items():
            for v in vs:                                                                    # every record lies inside a chunk of ITS bin
                assert any(c0 <= v < c1 for c0, c1 in got_bins[bn]), (tid, bn)
        n_intv = struct.unpack_from("<i", bai, off)[0]; off += 4
        lin = struct.unpack_from("<%dQ" % n_intv, bai, off); off += 8 * n_intv
        # 5.1.3: ioffset[w] = smallest virtual offset of a record overlapping 16 kb window w
        last_end = max(r["pos"] + sum(n for n, o in r["cigar"] if o in "MDN=X") for r in mine)
        assert n_intv == ((last_end - 1) >> 14) + 1
        for w in range(0, n_intv, max(1, n_intv // 64)):
            ov = [voff(r["data_off"]) for r in mine
                  if r["pos"] < (w + 1) << 14 and r["pos"] + sum(n for n, o in r["cigar"] if o in "MDN=X") > w << 14]
            if ov:
                assert lin[w] == min(ov), (tid, w)
    assert struct.unpack_from("<Q", bai, off)[0] == 2 and off + 8 == len(bai)               # n_no_coor


@pytest.mark.parametrize("bad", ["magic", "truncated_block", "crc"])
def test_reader_rejects_files_that_break_the_container_rules(tmp_path, bad):
    ex, _, _ = spec_records()
    path = str(tmp_path / "bad.bam")
    bam_file(path, "@HD\tVN:1.6\n", [("ref", 45)], ex)
    raw = bytearray(open(path, "rb").read())
    if bad == "magic":
        first = zlib.decompress(bytes(raw[18:struct.unpack_from("<H", raw, 16)[0] + 1 - 8]), -15)
        blk = bgzf_block(b"BAM\x02" + first[4:])
        raw = bytearray(blk) + raw[struct.unpack_from("<H", raw, 16)[0] + 1:]
    elif bad == "truncated_block":
        raw = raw[:len(raw) - 28 - 9]
    else:
        second = struct.unpack_from("<H", raw, 16)[0] + 1
        end = second + struct.unpack_from("<H", raw, second + 16)[0] + 1
        raw[end - 8] ^= 0x5A
    open(path, "wb").write(bytes(raw))
    with pytest.raises(Exception):
        with bamnative.NativeBamReader(path, threads=1) as rd:
            while rd.next_batch(4) is not None:
                pass
