"""ctypes binding of libccsm_bam (include/ccsm_bam.h): threaded BGZF/BAM reader that hands chunks of reads over in the layout
of ccsm_forward_reads_host, and the modbam writer (tag refill + MM/ML encoding + threaded BGZF).  The pure-Python
ccsmeth_amd/bamio.py implements the same formats record by record and is what the tests compare this library with."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libccsm_bam.so")

EXPORTS = ("ccsm_bam_last_error", "ccsm_bam_open", "ccsm_bam_header", "ccsm_bam_next", "ccsm_bam_batch_free", "ccsm_bam_close",
           "ccsm_bam_writer_open", "ccsm_bam_write_batch", "ccsm_bam_writer_flush", "ccsm_bam_writer_close",
           "ccsm_bam_modcalls_of_batch", "ccsm_bam_modcalls_free", "ccsm_bam_index_build", "ccsm_bam_sort",
           "ccsm_bam_align_info", "ccsm_bam_seek", "ccsm_bam_tell", "ccsm_bam_inflated_bytes", "ccsm_bam_seek_chunk", "ccsm_bam_eof_voffset",
           "ccsm_bam_writer_track_index", "ccsm_bam_writer_take_index", "ccsm_bam_index_write")


class _Batch(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("records", C.c_void_p), ("rec_offset", C.c_void_p), ("flag", C.c_void_p),
                ("offset", C.c_void_p), ("length", C.c_void_p), ("n_sites", C.c_void_p), ("seq", C.c_void_p), ("fi", C.c_void_p),
                ("ri", C.c_void_p), ("fp", C.c_void_p), ("rp", C.c_void_p), ("fn", C.c_void_p), ("rn", C.c_void_p),
                ("name_hash", C.c_void_p), ("total_bases", C.c_int64), ("voffset_start", C.c_uint64), ("voffset_end", C.c_uint64)]


class ModCallOpts(C.Structure):
    """ccsm_bam_modcall_opts: the per-record options of call_freqb (call_mods_freq_bam.py:486-537)."""
    _fields_ = [("identity", C.c_double), ("mapq", C.c_int32), ("no_supplementary", C.c_int32), ("base_clip", C.c_int32),
                ("refsites_all", C.c_int32), ("hap_tag", C.c_char * 2), ("modbase", C.c_char), ("modification", C.c_char)]


INDEX_ENTRY = np.dtype([("tid", "<i4"), ("pos", "<i4"), ("end", "<i4"), ("flag", "<u4"), ("vbeg", "<u8"), ("vend", "<u8")])


class _IndexRun(C.Structure):
    """ccsm_bam_index_run"""
    _fields_ = [("n_records", C.c_int64), ("n_unplaced", C.c_int64), ("sorted", C.c_int32), ("first_k1", C.c_uint64), ("first_k2", C.c_uint32),
                ("last_k1", C.c_uint64), ("last_k2", C.c_uint32), ("n_entries", C.c_int64), ("entries", C.c_void_p),
                ("file_start", C.c_int64), ("file_end", C.c_int64)]


class _ModCalls(C.Structure):
    _fields_ = [("n", C.c_int64), ("tid", C.c_void_p), ("pos", C.c_void_p), ("strand", C.c_void_p), ("ml", C.c_void_p),
                ("hap", C.c_void_p), ("n_records", C.c_int64), ("n_used", C.c_int64)]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libccsm_bam.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.ccsm_bam_last_error.restype = C.c_char_p
    lib.ccsm_bam_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    lib.ccsm_bam_header.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int64), C.POINTER(vp), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    lib.ccsm_bam_next.argtypes = [vp, C.c_int32, C.POINTER(C.POINTER(_Batch))]
    lib.ccsm_bam_batch_free.argtypes = [C.POINTER(_Batch)]
    lib.ccsm_bam_batch_free.restype = None
    lib.ccsm_bam_close.argtypes = [vp]
    lib.ccsm_bam_close.restype = None
    lib.ccsm_bam_writer_open.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, vp, C.c_int64, C.c_int32, C.c_int, C.c_int, C.POINTER(vp)]
    lib.ccsm_bam_write_batch.argtypes = [vp, C.POINTER(_Batch), vp, vp, vp, vp, C.c_int, C.POINTER(C.c_int32)]
    lib.ccsm_bam_writer_flush.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.ccsm_bam_writer_close.argtypes = [vp]
    lib.ccsm_bam_modcalls_of_batch.argtypes = [C.POINTER(_Batch), C.POINTER(ModCallOpts), vp, vp, C.c_int32, C.c_int,
                                               C.POINTER(C.POINTER(_ModCalls))]
    lib.ccsm_bam_modcalls_free.argtypes = [C.POINTER(_ModCalls)]
    lib.ccsm_bam_modcalls_free.restype = None
    lib.ccsm_bam_index_build.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
    lib.ccsm_bam_sort.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int64]
    lib.ccsm_bam_align_info.argtypes = [C.POINTER(_Batch), vp, vp, vp, vp]
    lib.ccsm_bam_seek.argtypes = [vp, C.c_uint64, C.c_uint64]
    lib.ccsm_bam_tell.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.ccsm_bam_inflated_bytes.argtypes = [vp]
    lib.ccsm_bam_inflated_bytes.restype = C.c_int64
    lib.ccsm_bam_seek_chunk.argtypes = [vp, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.ccsm_bam_eof_voffset.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.ccsm_bam_writer_track_index.argtypes = [vp, C.c_int]
    lib.ccsm_bam_writer_take_index.argtypes = [vp, C.POINTER(_IndexRun)]
    lib.ccsm_bam_index_write.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(_IndexRun), vp, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise IOError(load().ccsm_bam_last_error().decode("utf-8", "replace"))


def _view(ptr, dtype, count):
    if count == 0 or not ptr:
        return np.empty(0, dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(count,))


class Batch:
    """NumPy views of one ccsm_bam_batch (valid until close())."""

    def __init__(self, ptr):
        self._ptr = ptr
        b = ptr.contents
        n = self.n_reads = int(b.n_reads)
        self.total_bases = int(b.total_bases)
        self.rec_offset = _view(b.rec_offset, np.int64, n + 1)
        self.records = _view(b.records, np.uint8, int(self.rec_offset[-1]))
        self.flag = _view(b.flag, np.int32, n)
        self.offset = _view(b.offset, np.int64, n)
        self.length = _view(b.length, np.int32, n)
        self.n_sites = _view(b.n_sites, np.int32, n)
        self.seq, self.fi, self.ri, self.fp, self.rp = (_view(p, np.uint8, self.total_bases) for p in (b.seq, b.fi, b.ri, b.fp, b.rp))
        self.fn = _view(b.fn, np.float32, n)
        self.rn = _view(b.rn, np.float32, n)
        self.name_hash = _view(b.name_hash, np.uint64, n)
        self.voffset_start, self.voffset_end = int(b.voffset_start), int(b.voffset_end)

    def close(self):
        if self._ptr is not None:
            load().ccsm_bam_batch_free(self._ptr)
            self._ptr = None

    def __del__(self):
        self.close()


def modcalls_of_batch(batch, mapq=1, identity=0.0, no_supplementary=False, base_clip=0, refsites_all=False, hap_tag="HP",
                      modbase="C", modification="m", site_masks=None, threads=4):
    """Rows (tid, pos, strand, ml, hap) the records of `batch` contribute to call_freqb's per-position lists, as NumPy
    arrays (copies), plus (records seen, records used).  site_masks: list over reference ids of uint8 arrays (or None)."""
    if len(hap_tag) != 2:
        raise ValueError("--hap_tag must be a two-character tag")
    o = ModCallOpts(float(identity), int(mapq), int(bool(no_supplementary)), int(base_clip), int(bool(refsites_all)),
                    hap_tag.encode("ascii"), modbase.encode("ascii"), modification.encode("ascii"))
    n_ref = len(site_masks) if site_masks is not None else 0
    ptrs = lens = None
    if site_masks is not None:
        keep = [None if m is None else np.ascontiguousarray(m, dtype=np.uint8) for m in site_masks]
        ptrs = (C.c_void_p * max(n_ref, 1))(*[None if m is None else m.ctypes.data for m in keep])
        lens = (C.c_int64 * max(n_ref, 1))(*[0 if m is None else len(m) for m in keep])
    p = C.POINTER(_ModCalls)()
    _check(load().ccsm_bam_modcalls_of_batch(batch._ptr, C.byref(o), ptrs, lens, n_ref, int(threads), C.byref(p)))
    try:
        c = p.contents
        n = int(c.n)
        out = tuple(_view(q, dt, n).copy() for q, dt in ((c.tid, np.int32), (c.pos, np.int32), (c.strand, np.uint8),
                                                         (c.ml, np.uint8), (c.hap, np.uint8)))
        return out + (int(c.n_records), int(c.n_used))
    finally:
        _lib.ccsm_bam_modcalls_free(p)


class NativeBamReader:
    def __init__(self, path, threads=4):
        self._h = C.c_void_p()
        _check(load().ccsm_bam_open(os.fsencode(path), int(threads), C.byref(self._h)))
        text, refs = C.c_void_p(), C.c_void_p()
        tl, rl, nref = C.c_int64(), C.c_int64(), C.c_int32()
        _check(_lib.ccsm_bam_header(self._h, C.byref(text), C.byref(tl), C.byref(refs), C.byref(rl), C.byref(nref)))
        self.header_text = C.string_at(text, tl.value).decode("utf-8", "replace") if tl.value else ""
        self.raw_refs = C.string_at(refs, rl.value) if rl.value else b""
        self.n_ref = nref.value

    def next_batch(self, max_reads):
        """Batch of up to max_reads records, or None at end of file."""
        p = C.POINTER(_Batch)()
        _check(_lib.ccsm_bam_next(self._h, int(max_reads), C.byref(p)))
        return Batch(p) if p else None

    def seek(self, voffset_start, voffset_end=0):
        """Continue at a BGZF virtual offset; with voffset_end the range ends like a file (nothing behind its block is inflated)."""
        _check(_lib.ccsm_bam_seek(self._h, int(voffset_start), int(voffset_end)))

    def seek_chunk(self, coffset_lo, coffset_hi):
        """Position at the first record that starts in a BGZF block of file range [coffset_lo, coffset_hi); next_batch then returns
        the records starting there and None behind them.  -> that record's virtual offset, or 0 when the chunk holds no record start."""
        v = C.c_uint64(0)
        _check(_lib.ccsm_bam_seek_chunk(self._h, int(coffset_lo), int(coffset_hi), C.byref(v)))
        return v.value

    def tell(self):
        v = C.c_uint64(0)
        _check(_lib.ccsm_bam_tell(self._h, C.byref(v)))
        return v.value

    def eof_voffset(self):
        """What tell() reports behind the file's last record, from the BGZF block headers alone (the hand-over chain must end here)."""
        v = C.c_uint64(0)
        _check(_lib.ccsm_bam_eof_voffset(self._h, C.byref(v)))
        return v.value

    @property
    def inflated_bytes(self):
        return int(_lib.ccsm_bam_inflated_bytes(self._h))

    def close(self):
        if self._h:
            _lib.ccsm_bam_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class NativeBamWriter:
    def __init__(self, path, header_text, raw_refs=b"", n_ref=0, threads=4, level=6):
        self._h = C.c_void_p()
        text = header_text.encode("utf-8")
        self._refs = raw_refs
        _check(load().ccsm_bam_writer_open(os.fsencode(path), text, len(text), C.cast(C.c_char_p(raw_refs), C.c_void_p) if raw_refs else None,
                                           len(raw_refs), int(n_ref), int(threads), int(level), C.byref(self._h)))

    def write_batch(self, batch, first_site=None, locs=None, prob1=None, tagged=None, rm_pulse=True):
        """Write every record of `batch`; returns the number of reads that received MM/ML."""
        keep = []

        def ptr(a, dt):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dt)
            keep.append(a)
            return a.ctypes.data
        n = C.c_int32(0)
        _check(_lib.ccsm_bam_write_batch(self._h, batch._ptr, ptr(first_site, np.int32), ptr(locs, np.int32), ptr(prob1, np.float32),
                                         ptr(tagged, np.uint8), int(bool(rm_pulse)), C.byref(n)))
        return n.value

    def flush(self):
        """End the current BGZF block (= the current run); returns the file size so far."""
        off = C.c_int64(0)
        _check(_lib.ccsm_bam_writer_flush(self._h, C.byref(off)))
        return off.value

    def track_index(self, enable=True):
        """Keep the index bookkeeping of the records written from here on (call right after a flush)."""
        _check(_lib.ccsm_bam_writer_track_index(self._h, int(bool(enable))))

    def take_index(self):
        """IndexRun of everything written since the previous take; call right after flush()."""
        r = _IndexRun()
        _check(_lib.ccsm_bam_writer_take_index(self._h, C.byref(r)))
        ent = np.empty(0, INDEX_ENTRY)
        if r.n_entries:
            ent = np.ctypeslib.as_array(C.cast(r.entries, C.POINTER(C.c_uint8)), shape=(int(r.n_entries) * INDEX_ENTRY.itemsize,)).view(INDEX_ENTRY).copy()
        return IndexRun(int(r.n_records), int(r.n_unplaced), bool(r.sorted), (int(r.first_k1), int(r.first_k2)), (int(r.last_k1), int(r.last_k2)),
                        ent, int(r.file_start), int(r.file_end))

    def close(self):
        if self._h:
            h, self._h = self._h, C.c_void_p()
            _check(_lib.ccsm_bam_writer_close(h))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class IndexRun:
    """One run table of NativeBamWriter.take_index (include/ccsm_bam.h: ccsm_bam_index_run); picklable."""
    __slots__ = ("n_records", "n_unplaced", "sorted", "first_key", "last_key", "entries", "file_start", "file_end")

    def __init__(self, n_records, n_unplaced, is_sorted, first_key, last_key, entries, file_start, file_end):
        self.n_records, self.n_unplaced, self.sorted, self.first_key, self.last_key = n_records, n_unplaced, is_sorted, first_key, last_key
        self.entries, self.file_start, self.file_end = entries, file_start, file_end

    def __getstate__(self):
        return tuple(getattr(self, k) for k in self.__slots__)

    def __setstate__(self, st):
        for k, v in zip(self.__slots__, st):
            setattr(self, k, v)

    def to_wire(self):
        """Plain-data form for the ranks' gather over the store (sharding.ChunkQueue.rendezvous: JSON, nothing executable)."""
        import base64
        return {"__index_run__": [self.n_records, self.n_unplaced, bool(self.sorted), list(self.first_key), list(self.last_key),
                                  base64.b64encode(np.ascontiguousarray(self.entries, INDEX_ENTRY).tobytes()).decode("ascii"),
                                  self.file_start, self.file_end]}

    @classmethod
    def from_wire(cls, d):
        import base64
        nr, nu, srt, fk, lk, ent, fs, fe = d["__index_run__"]
        return cls(int(nr), int(nu), bool(srt), (int(fk[0]), int(fk[1])), (int(lk[0]), int(lk[1])),
                   np.frombuffer(base64.b64decode(ent), INDEX_ENTRY).copy(), int(fs), int(fe))


def index_write(bai_path, n_ref, runs, shifts=None):
    """Write <bai_path> from IndexRun tables given in final file order (shifts[i] = how far run i's bytes were moved when the part
    files were stitched).  -> (sorted, n_records); nothing is written when the records are not in coordinate order."""
    n = len(runs)
    arr = (_IndexRun * max(n, 1))()
    keep = []
    for i, r in enumerate(runs):
        ent = np.ascontiguousarray(r.entries, INDEX_ENTRY)
        keep.append(ent)
        arr[i] = _IndexRun(r.n_records, r.n_unplaced, int(r.sorted), r.first_key[0], r.first_key[1], r.last_key[0], r.last_key[1], len(ent),
                           ent.ctypes.data if len(ent) else None, r.file_start, r.file_end)
    sh = np.ascontiguousarray(shifts if shifts is not None else np.zeros(n), np.int64)
    srt, cnt = C.c_int(), C.c_int64()
    _check(load().ccsm_bam_index_write(os.fsencode(bai_path), int(n_ref), n, arr, sh.ctypes.data if n else None, C.byref(srt), C.byref(cnt)))
    return bool(srt.value), int(cnt.value)


BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def stitch_runs(out_path, header_file, header_end, runs):
    """Multi-GPU merge: `header_file`[0:header_end] (the block-aligned BAM header), then every (file, start, end) run of BGZF
    blocks in the given order, then the BGZF end-of-file marker."""
    with open(out_path, "wb") as out:
        with open(header_file, "rb") as fh:
            out.write(fh.read(header_end))
        handles = {}
        try:
            for path, start, end in runs:
                fh = handles.get(path)
                if fh is None:
                    fh = handles[path] = open(path, "rb")
                fh.seek(start)
                left = end - start
                while left > 0:
                    chunk = fh.read(min(left, 1 << 24))
                    if not chunk:
                        raise IOError("short read while stitching %s" % path)
                    out.write(chunk)
                    left -= len(chunk)
        finally:
            for fh in handles.values():
                fh.close()
        out.write(BGZF_EOF)


def stitch_layout(header_end, spans):
    """Final file offsets of the runs [(path, start, end)] laid back to back behind the header, and the file's total size."""
    dst, off = [], int(header_end)
    for _, a, e in spans:
        dst.append(off)
        off += int(e) - int(a)
    return dst, off + len(BGZF_EOF)


def stitch_create(out_path, header_file, header_end, total_size):
    """The output file with its header blocks in place, sized for every run, the BGZF end-of-file block at its end."""
    with open(out_path, "wb") as out:
        with open(header_file, "rb") as fh:
            out.write(fh.read(header_end))
        out.truncate(total_size)
        out.seek(total_size - len(BGZF_EOF))
        out.write(BGZF_EOF)


def stitch_copy(out_path, spans, dst):
    """Copy the runs [(path, start, end)] to their offsets `dst` of the (existing, sized) output: in the kernel where the file
    system allows it (copy_file_range), else through a buffer.  Every rank copies its own runs at the same time; no byte is
    touched twice and nothing is recompressed."""
    fds = {}
    out = os.open(out_path, os.O_WRONLY)
    use_cfr = hasattr(os, "copy_file_range")
    try:
        for (path, start, end), d in zip(spans, dst):
            fd = fds.get(path)
            if fd is None:
                fd = fds[path] = os.open(path, os.O_RDONLY)
            src, left = int(start), int(end) - int(start)
            while left > 0:
                n = 0
                if use_cfr:
                    try:
                        n = os.copy_file_range(fd, out, min(left, 1 << 30), src, d)
                    except OSError:
                        use_cfr = False
                if not use_cfr:
                    buf = os.pread(fd, min(left, 1 << 24), src)
                    if not buf:
                        raise IOError("short read while stitching %s" % path)
                    n = os.pwrite(out, buf, d)
                elif n == 0:
                    raise IOError("short read while stitching %s" % path)
                src += n
                d += n
                left -= n
    finally:
        os.close(out)
        for fd in fds.values():
            os.close(fd)


def stitch_runs_parallel(out_path, header_file, header_end, spans, mine, finish):
    """Multi-GPU merge, every rank's share: `spans` = all runs in final order, `mine` = the indices this caller copies, finish = this
    caller also creates the file (it must do so before anybody copies: the callers synchronise around it).  -> destination offsets."""
    dst, total = stitch_layout(header_end, spans)
    if finish:
        stitch_create(out_path, header_file, header_end, total)
    mine = list(mine)
    stitch_copy(out_path, [spans[i] for i in mine], [dst[i] for i in mine])
    return dst


def align_info(batch):
    """-> (mapq int32, query_alignment_start int32, query_alignment_end int32, identity float64) per record of the batch."""
    n = batch.n_reads
    mapq, qs, qe, ident = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.float64)
    _check(load().ccsm_bam_align_info(batch._ptr, mapq.ctypes.data, qs.ctypes.data, qe.ctypes.data, ident.ctypes.data))
    return mapq, qs, qe, ident


def index_build(bam_path, bai_path=None, threads=4):
    """One streaming pass: -> (sorted, n_records); writes <bam>.bai when the records are in coordinate order."""
    srt, n = C.c_int(), C.c_int64()
    _check(load().ccsm_bam_index_build(os.fsencode(bam_path), os.fsencode(bai_path or bam_path + ".bai"), int(threads), C.byref(srt), C.byref(n)))
    return bool(srt.value), int(n.value)


def sort_and_index(bam_path, threads=4, level=6, max_bytes=0):
    """The reference's post-processing (call_modifications.py:592-607: samtools sort -o x.sorted.bam; rename; samtools index):
    index in place when the file is already in coordinate order, else sort it (in memory) first.  -> True when a sort ran."""
    ok, _ = index_build(bam_path, threads=threads)
    if ok:
        return False
    tmp = os.path.splitext(bam_path)[0] + ".sorted.bam"
    _check(load().ccsm_bam_sort(os.fsencode(bam_path), os.fsencode(tmp), int(threads), int(level), int(max_bytes)))
    os.replace(tmp, bam_path)
    ok, _ = index_build(bam_path, threads=threads)
    if not ok:
        raise IOError("sorted file is not in coordinate order")
    return True
