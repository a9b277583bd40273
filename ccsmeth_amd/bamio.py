"""Minimal BGZF / BAM reader and writer (no htslib, no pysam) for the call_mods path.

Replaces what reference call_modifications.py / extract_features.py do through pysam on this path
(extract_features.py:88-126 tag access, :129-177 reader; call_modifications.py:410-462 writer, _bam2modbam.py:211-226
tag refill): sequential read of unaligned or aligned HiFi BAM records with their kinetics tags (fi/ri/fp/rp B:C, fn/rn
integers, sn B:f), and sequential write of the same records with MM/ML replaced.  Follows the SAM/BAM specification v1
(BGZF blocks = gzip members with a 'BC' extra field; little-endian BAM records).  Sorting/indexing (pysam.sort/index,
call_modifications.py:592-607) is not provided: output order = input order, i.e. the reference's `--no_sort` behaviour.
"""
import struct
import zlib

import numpy as np

_SEQ_DECODE = np.frombuffer(b"=ACMGRSVTWYHKDBN", dtype=np.uint8)
_SEQ_ENCODE = np.full(256, 15, dtype=np.uint8)
for _i, _c in enumerate(b"=ACMGRSVTWYHKDBN"):
    _SEQ_ENCODE[_c] = _i
    _SEQ_ENCODE[ord(chr(_c).lower())] = _i
_COMP = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _COMP[_a] = _b
_B_DTYPES = {"c": np.int8, "C": np.uint8, "s": np.dtype("<i2"), "S": np.dtype("<u2"), "i": np.dtype("<i4"),
             "I": np.dtype("<u4"), "f": np.dtype("<f4")}
_SCALAR = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f"}
_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
_BLOCK_PAYLOAD = 0xff00


# ---------------------------------------------------------------------------------------------------------------------
# BGZF
# ---------------------------------------------------------------------------------------------------------------------
def bgzf_blocks(fh):
    """Yield the decompressed payload of every BGZF block of an open binary file."""
    while True:
        head = fh.read(12)
        if len(head) == 0:
            return
        if len(head) < 12 or head[:4] != b"\x1f\x8b\x08\x04":
            raise ValueError("not a BGZF stream (bad gzip member header)")
        xlen = struct.unpack_from("<H", head, 10)[0]
        extra = fh.read(xlen)
        bsize = None
        off = 0
        while off + 4 <= xlen:
            si1, si2, slen = extra[off], extra[off + 1], struct.unpack_from("<H", extra, off + 2)[0]
            if si1 == 66 and si2 == 67 and slen == 2:
                bsize = struct.unpack_from("<H", extra, off + 4)[0]
            off += 4 + slen
        if bsize is None:
            raise ValueError("gzip member without the BGZF 'BC' field")
        cdata_len = bsize - xlen - 19
        cdata = fh.read(cdata_len)
        crc, isize = struct.unpack("<II", fh.read(8))
        data = zlib.decompress(cdata, -15) if isize else b""
        if len(data) != isize or (zlib.crc32(data) & 0xffffffff) != crc:
            raise ValueError("BGZF block failed its CRC / size check")
        yield data


def bgzf_compress_block(data, level=6):
    assert len(data) <= _BLOCK_PAYLOAD
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    cdata = co.compress(data) + co.flush()
    bsize = len(cdata) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + cdata +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


class _ByteStream:
    """Sequential reader over the concatenated BGZF payloads."""

    def __init__(self, fh):
        self._it = bgzf_blocks(fh)
        self._buf = b""
        self._pos = 0

    def read(self, n):
        while len(self._buf) - self._pos < n:
            try:
                nxt = next(self._it)
            except StopIteration:
                break
            self._buf = self._buf[self._pos:] + nxt
            self._pos = 0
        out = self._buf[self._pos:self._pos + n]
        self._pos += len(out)
        return out


# ---------------------------------------------------------------------------------------------------------------------
# Records
# ---------------------------------------------------------------------------------------------------------------------
class BamRecord:
    """One alignment record.  `tags` = list of (tag, type, value): type is the BAM type char ('A','c','C','s','S','i','I',
    'f','Z','H') or 'B<sub>' for arrays (value = NumPy array)."""

    __slots__ = ("query_name", "flag", "ref_id", "pos", "mapq", "cigar", "next_ref_id", "next_pos", "tlen", "seq", "qual",
                 "tags")

    def __init__(self, query_name, flag=4, ref_id=-1, pos=-1, mapq=255, cigar=(), next_ref_id=-1, next_pos=-1, tlen=0,
                 seq="", qual=None, tags=None):
        self.query_name, self.flag, self.ref_id, self.pos, self.mapq = query_name, flag, ref_id, pos, mapq
        self.cigar, self.next_ref_id, self.next_pos, self.tlen = tuple(cigar), next_ref_id, next_pos, tlen
        self.seq, self.qual, self.tags = seq, qual, list(tags or [])

    @property
    def is_reverse(self):
        return bool(self.flag & 16)

    @property
    def is_unmapped(self):
        return bool(self.flag & 4)

    def get_forward_sequence(self):
        """Sequence as it came off the instrument (reverse-complemented back when the record is reverse-strand)."""
        if not self.is_reverse:
            return self.seq
        b = np.frombuffer(self.seq.encode("ascii"), dtype=np.uint8)
        return _COMP[b[::-1]].tobytes().decode("ascii")

    def has_tag(self, tag):
        return any(t[0] == tag for t in self.tags)

    def get_tag(self, tag):
        for t, _, v in self.tags:
            if t == tag:
                return v
        raise KeyError(tag)


def _parse_tags(buf, off, end):
    tags = []
    while off < end:
        tag = buf[off:off + 2].decode("ascii")
        typ = chr(buf[off + 2])
        off += 3
        if typ == "A":
            val = chr(buf[off]); off += 1
        elif typ in _SCALAR:
            fmt = _SCALAR[typ]
            val = struct.unpack_from(fmt, buf, off)[0]
            off += struct.calcsize(fmt)
        elif typ in "ZH":
            z = buf.index(b"\x00", off)
            val = buf[off:z].decode("ascii")
            off = z + 1
        elif typ == "B":
            sub = chr(buf[off])
            cnt = struct.unpack_from("<i", buf, off + 1)[0]
            dt = np.dtype(_B_DTYPES[sub])
            val = np.frombuffer(buf, dtype=dt, count=cnt, offset=off + 5).copy()
            off += 5 + cnt * dt.itemsize
            typ = "B" + sub
        else:
            raise ValueError("unknown BAM tag type %r" % typ)
        tags.append((tag, typ, val))
    return tags


def _encode_tags(tags):
    out = bytearray()
    for tag, typ, val in tags:
        out += tag.encode("ascii")
        if typ == "A":
            out += b"A" + val.encode("ascii")
        elif typ in _SCALAR:
            out += typ.encode("ascii") + struct.pack(_SCALAR[typ], val)
        elif typ in ("Z", "H"):
            out += typ.encode("ascii") + val.encode("ascii") + b"\x00"
        elif typ[0] == "B":
            arr = np.ascontiguousarray(val, dtype=_B_DTYPES[typ[1]])
            out += b"B" + typ[1].encode("ascii") + struct.pack("<i", arr.size) + arr.tobytes()
        else:
            raise ValueError("unknown BAM tag type %r" % typ)
    return bytes(out)


def int_tag_type(v):
    """Smallest BAM integer type holding v (what htslib/pysam choose when a tag is set from a Python int)."""
    if v >= 0:
        return "C" if v <= 0xff else ("S" if v <= 0xffff else "I")
    return "c" if v >= -128 else ("s" if v >= -32768 else "i")


def _reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


class BamReader:
    """Sequential BAM reader: `header_text`, `references` [(name, length)], iteration yields BamRecord."""

    def __init__(self, path):
        self._fh = open(path, "rb")
        self._s = _ByteStream(self._fh)
        if self._s.read(4) != b"BAM\x01":
            raise ValueError("%s is not a BAM file" % path)
        l_text = struct.unpack("<i", self._s.read(4))[0]
        self.header_text = self._s.read(l_text).split(b"\x00", 1)[0].decode("utf-8")
        n_ref = struct.unpack("<i", self._s.read(4))[0]
        self.references = []
        for _ in range(n_ref):
            l_name = struct.unpack("<i", self._s.read(4))[0]
            name = self._s.read(l_name)[:-1].decode("ascii")
            self.references.append((name, struct.unpack("<i", self._s.read(4))[0]))

    def close(self):
        self._fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __iter__(self):
        return self

    def __next__(self):
        head = self._s.read(4)
        if len(head) < 4:
            raise StopIteration
        size = struct.unpack("<i", head)[0]
        buf = self._s.read(size)
        if len(buf) < size:
            raise ValueError("truncated BAM record")
        ref_id, pos, l_name, mapq, _bin, n_cig, flag, l_seq, nref, npos, tlen = struct.unpack_from("<iiBBHHHiiii", buf, 0)
        off = 32
        name = buf[off:off + l_name - 1].decode("ascii")
        off += l_name
        cig = np.frombuffer(buf, dtype="<u4", count=n_cig, offset=off)
        cigar = tuple((int(c & 0xf), int(c >> 4)) for c in cig)
        off += 4 * n_cig
        packed = np.frombuffer(buf, dtype=np.uint8, count=(l_seq + 1) // 2, offset=off)
        nib = np.empty(2 * len(packed), dtype=np.uint8)
        nib[0::2] = packed >> 4
        nib[1::2] = packed & 0xf
        seq = _SEQ_DECODE[nib[:l_seq]].tobytes().decode("ascii")
        off += (l_seq + 1) // 2
        q = np.frombuffer(buf, dtype=np.uint8, count=l_seq, offset=off).copy()
        qual = None if (l_seq > 0 and q[0] == 0xff) else q
        off += l_seq
        return BamRecord(name, flag, ref_id, pos, mapq, cigar, nref, npos, tlen, seq, qual, _parse_tags(buf, off, size))


class BamWriter:
    def __init__(self, path, header_text, references=(), level=6):
        self._fh = open(path, "wb")
        self._level = level
        self._buf = bytearray()
        text = header_text.encode("utf-8")
        self._put(b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(references)))
        for name, length in references:
            nm = name.encode("ascii") + b"\x00"
            self._put(struct.pack("<i", len(nm)) + nm + struct.pack("<i", length))

    def _put(self, data):
        self._buf += data
        while len(self._buf) >= _BLOCK_PAYLOAD:
            self._fh.write(bgzf_compress_block(bytes(self._buf[:_BLOCK_PAYLOAD]), self._level))
            del self._buf[:_BLOCK_PAYLOAD]

    def write(self, rec):
        name = rec.query_name.encode("ascii") + b"\x00"
        l_seq = len(rec.seq)
        codes = _SEQ_ENCODE[np.frombuffer(rec.seq.encode("ascii"), dtype=np.uint8)]
        if l_seq & 1:
            codes = np.append(codes, 0)
        packed = ((codes[0::2] << 4) | codes[1::2]).astype(np.uint8).tobytes()
        qual = (np.full(l_seq, 0xff, np.uint8) if rec.qual is None else np.asarray(rec.qual, np.uint8)).tobytes()
        cigar = np.array([(ln << 4) | op for op, ln in rec.cigar], dtype="<u4").tobytes()
        ref_len = sum(ln for op, ln in rec.cigar if op in (0, 2, 3, 7, 8))
        end = rec.pos + (ref_len if ref_len > 0 else 1)
        bin_ = _reg2bin(rec.pos, end) if rec.pos >= 0 else 4680
        body = (struct.pack("<iiBBHHHiiii", rec.ref_id, rec.pos, len(name), rec.mapq, bin_, len(rec.cigar), rec.flag, l_seq,
                            rec.next_ref_id, rec.next_pos, rec.tlen) + name + cigar + packed + qual + _encode_tags(rec.tags))
        self._put(struct.pack("<i", len(body)) + body)

    def close(self):
        if self._fh is None:
            return
        if self._buf:
            self._fh.write(bgzf_compress_block(bytes(self._buf), self._level))
            self._buf = bytearray()
        self._fh.write(_BGZF_EOF)
        self._fh.close()
        self._fh = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def add_pg_line(header_text, version, command_line):
    """call_modifications.py:445: append an @PG record {PN ccsmeth, ID ccsmeth, VN, CL}."""
    line = "@PG\tID:ccsmeth\tPN:ccsmeth\tVN:%s\tCL:%s" % (version, command_line)
    if header_text and not header_text.endswith("\n"):
        header_text += "\n"
    return header_text + line + "\n"
