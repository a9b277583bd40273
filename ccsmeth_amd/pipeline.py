"""Per-read call_mods pipeline on one GPU: reads -> 21-mer features (host, vectorised) -> libccsm forward through the
pinned double-buffered staging ring (ccsm_submit_host / ccsm_wait_host on two workspaces and two HIP streams) ->
per-site round(p1/(p0+p1), 6) -> MM / ML tags per read.

Mirrors what reference call_modifications.py does between its reader and writer processes
(process_one_holebatch :extract_features.py:409-431, _call_mods2s :170-227, _add_modinfo2alignedseg :230-263) for
denovo mode, without pysam: a read is (name, forward_seq, fi, ri, fp, rp, fn, rn, is_reverse).  BAM parsing/writing
(a-1, a-10) is outside this round."""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import _lib
from ._bam2modbam import _convert_locs_to_mmtag, _convert_probs_to_mltag
from .call_modifications import prob1_norm_round6
from .extract_features import count_kept_sites, extract_read_arrays

Read = namedtuple("Read", "name seq fi ri fp rp fn rn is_reverse")


class ArithmeticViolation(RuntimeError):
    """The running split3 shadow found split-mx outside the selection rule later in the input (Pipeline.shadow_every)."""


def read_key(name):
    """64-bit FNV-1a of the read name: the key of a read's device-drawn initial states (ccsm_reads.h0_key; the native reader
    hands the same value over as ccsm_bam_batch.name_hash).  A site draws from (this key, position of its C in the read), so its
    probability does not depend on where the read stands in the file, on the batching or on which GPU handles it."""
    h = 0xcbf29ce484222325
    for c in (name.encode("ascii", "replace") if isinstance(name, str) else bytes(name)):
        h = ((h ^ c) * 0x100000001b3) & 0xffffffffffffffff
    return h
ReadCalls = namedtuple("ReadCalls", "name n_sites locs probs mm ml mm_flag")


class _Slot:
    """One workspace + its pending host-side bookkeeping."""

    def __init__(self, dm, max_sites, stream):
        self.ws = dm.workspace(max_sites)
        self.stream = stream
        self.pending = None


class CallModsPipeline:
    def __init__(self, device_model, batch_size=2048, seed=1234, extract="host", norm="zscore", no_decode=False):
        """extract="host": NumPy feature extraction + feature-level C-ABI (ccsm_submit_host / ccsm_wait_host);
        extract="device": raw read arrays go to the GPU, ccsm_forward_reads_host extracts there (include/ccsm.h)."""
        import torch
        if extract not in ("host", "device"):
            raise ValueError("extract must be 'host' or 'device'")
        self.dm = device_model
        self.batch_size = int(batch_size)
        self.seed = seed
        self.extract = extract
        self.norm, self.no_decode = norm, bool(no_decode)       # host extraction only (the device kernels: zscore on decoded codes)
        if extract == "device" and (norm != "zscore" or no_decode):
            raise ValueError("the device extraction kernels implement --norm zscore with CodecV1 decoding; use extract='host'")
        self._rws = None
        self._rwss = [None, None]                       # read-level workspaces of the double-buffered native path
        dev = torch.device("cuda", device_model.device)
        self._streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
        self._slots = [_Slot(device_model, self.batch_size, s.cuda_stream) for s in self._streams]
        self._inflight = []                             # [(workspace slot, job, selection, shadowed)] of the native-batch path, oldest first
        self._turn = 0
        # The running split3 shadow (call_mods, split-mx behind both probes): one chunk in `shadow_every` is ALSO run in split3, on a workspace
        # of its own behind the chunk's launch, and compared when the chunk is collected; a chunk beyond `shadow_limit` raises
        # ArithmeticViolation (the caller restarts in split3).  0 = off
        self.raw_log = None                             # a list: run() also appends every device batch's raw (n, 2) probabilities (call_mods' probe on the record-level path)
        self.shadow_every, self.shadow_limit = 0, 1.25e-5
        self._shadow_ws = None
        self.shadow_chunks, self.shadow_sites, self.shadow_max = 0, 0, -1.0

    def close(self, discard=False):
        """discard: drop what is still in flight (an aborted run) instead of collecting - and checking - it"""
        while self._inflight:
            if discard:
                k, _, _, shadowed = self._inflight.pop(0)
                self._rwss[k].wait_reads()
                if shadowed:
                    self._shadow_ws.wait_reads()
            else:
                self._collect_one()
        if self._shadow_ws is not None:
            self._shadow_ws.close()
            self._shadow_ws = None
        for s in self._slots:
            s.ws.close()
        if self._rws is not None:
            self._rws.close()
        for w in self._rwss:
            if w is not None:
                w.close()

    # ---- device-side extraction -----------------------------------------------------------------------------------
    def _run_device(self, reads):
        """Chunks of whole reads holding <= batch_size sites (one larger read alone if it exceeds that) go through
        ccsm_forward_reads_host.  The Philox counter of a site's initial states is (hash of the read name, position of the C), as in
        the host path and the native-BAM path, so all three give the same probabilities for the same seed."""
        out = [None] * len(reads)
        failed = 0
        cnts = []
        for r in reads:
            ok = all(len(a) == len(r.seq) for a in (r.fi, r.ri, r.fp, r.rp))          # extract_features.py:320-325
            cnts.append(count_kept_sites(np.frombuffer(r.seq.encode("ascii"), np.uint8)) if ok else 0)
        chunk, csites = [], 0

        def flush():
            nonlocal chunk, csites
            if not chunk:
                return
            if self._rws is None or self._rws.max_sites < csites:
                if self._rws is not None:
                    self._rws.close()
                self._rws = self.dm.workspace(max(csites, self.batch_size))
            rd = [(reads[i].seq, reads[i].fi, reads[i].ri, reads[i].fp, reads[i].rp, reads[i].fn, reads[i].rn) for i in chunk]
            first, locs, _, probs = self._rws.forward_reads(rd, seed=self.seed, stream=self._slots[0].stream,
                                                           read_keys=np.array([read_key(reads[i].name) for i in chunk], np.uint64))
            if self.raw_log is not None:
                self.raw_log.append(np.array(probs, np.float32, copy=True))
            p1 = prob1_norm_round6(probs)
            for j, i in enumerate(chunk):
                a, b = int(first[j]), int(first[j + 1])
                out[i] = _tags_for_read(reads[i], locs[a:b].astype(np.int64), p1[a:b])
            chunk, csites = [], 0

        for i, (r, c) in enumerate(zip(reads, cnts)):
            if c == 0:
                failed += 1
                out[i] = ReadCalls(r.name, 0, np.empty(0, np.int64), np.empty(0, np.float32), None, None, 0)
                continue
            if csites + c > self.batch_size:
                flush()
            chunk.append(i)
            csites += c
        flush()
        return out, failed

    # ---- device side ---------------------------------------------------------------------------------------------
    def _submit(self, slot, feats, meta):
        n = feats["kmer1"].shape[0]
        lib = self.dm._lib
        b = _lib.Batch()
        keep = []
        for s, sfx in enumerate(("1", "2")):
            arrs = [np.ascontiguousarray(feats["kmer" + sfx], np.uint8), np.ascontiguousarray(feats["ipd" + sfx], np.float32),
                    np.ascontiguousarray(feats["pw" + sfx], np.float32), np.ascontiguousarray(feats["npass" + sfx], np.float32)]
            keep += arrs
            b.strand[s].kmer, b.strand[s].ipd, b.strand[s].pw, b.strand[s].npass = (a.ctypes.data for a in arrs)
        b.kmer_is_f32, b.npass_per_base = 0, 0
        h = _lib.H0()
        h.mode, h.seed = _lib.H0_DEVICE_RNG, self.seed
        skey = np.ascontiguousarray(feats["site_key"], np.uint64)
        ssub = np.ascontiguousarray(feats["site_sub"], np.uint32)
        keep += [skey, ssub]
        h.site_key, h.site_sub = skey.ctypes.data, ssub.ctypes.data
        _lib.check(lib.ccsm_submit_host(self.dm.handle, slot.ws.handle, n, C.byref(b), C.byref(h), slot.stream))
        slot.pending = (n, meta)

    def _collect(self, slot, acc):
        if slot.pending is None:
            return
        n, meta = slot.pending
        slot.pending = None
        logits = np.empty((n, 2), np.float32)
        probs = np.empty((n, 2), np.float32)
        _lib.check(self.dm._lib.ccsm_wait_host(slot.ws.handle, logits.ctypes.data, probs.ctypes.data))
        if self.raw_log is not None:
            self.raw_log.append(probs.copy())
        p1 = prob1_norm_round6(probs)
        pos = 0
        for ridx, locs in meta:
            k = len(locs)
            acc[ridx].append((locs, p1[pos:pos + k]))
            pos += k

    # ---- native BAM batches (ccsmeth_amd/bamnative.py): no per-read Python objects ------------------------------------------
    def feed_native_batch(self, batch, skip=None):
        """Queue one bamnative.Batch: chunks of whole reads holding <= batch_size sites go through ccsm_submit_reads_host, double-buffered
        over two workspaces and two streams (the reader's site counts make the submit non-blocking).  At most two chunks are in flight;
        submitting a third first collects the oldest, which may belong to the PREVIOUS batch: the GPU does not drain between
        hole-batches.  Returns a job for finish_native_batch; `batch` may be released once this returns (submit copies the arrays)."""
        nr = batch.n_reads
        cnt = np.where(batch.length > 0, batch.n_sites, 0).astype(np.int64)
        if skip is not None:                            # reads excluded by name (--holeids_e / --holeids_ne): no features
            cnt[np.asarray(skip, bool)] = 0
        first = np.zeros(nr + 1, np.int32)
        np.cumsum(cnt, out=first[1:])
        total = int(first[-1])
        job = dict(first=first, cnt=cnt, locs=np.empty(total, np.int32), prob1=np.empty(total, np.float32), tagged=(cnt > 0).astype(np.uint8),
                   pending=0, failed=int(nr - np.count_nonzero(cnt > 0)))
        idx = np.flatnonzero(cnt > 0)
        start = 0
        while start < len(idx):
            csum = np.cumsum(cnt[idx[start:]])
            take = max(1, int(np.searchsorted(csum, self.batch_size, side="right")))
            sel = idx[start:start + take]
            csites = int(cnt[sel].sum())
            k = self._turn & 1
            if len(self._inflight) == 2:
                self._collect_one()
            if self._rwss[k] is None or self._rwss[k].max_sites < csites:
                if self._rwss[k] is not None:
                    self._rwss[k].close()
                self._rwss[k] = self.dm.workspace(max(csites, self.batch_size))
            self._rwss[k].submit_reads_arrays(batch.offset[sel], batch.length[sel], batch.seq, batch.fi, batch.ri, batch.fp, batch.rp,
                                              batch.fn[sel], batch.rn[sel], site_counts=cnt[sel], seed=self.seed,
                                              stream=self._slots[k].stream, read_keys=batch.name_hash[sel])
            shadowed = False
            if self.shadow_every and self._shadow_ws_free() and self._turn % self.shadow_every == self.shadow_every - 1:
                # the same chunk again in split3 (ccsm_workspace_force_split3: the model itself keeps serving split-mx), same stream, same keyed
                # initial states: compared at collection
                if self._shadow_ws is None or self._shadow_ws.max_sites < csites:
                    if self._shadow_ws is not None:
                        self._shadow_ws.close()
                    self._shadow_ws = self.dm.workspace(max(csites, self.batch_size))
                self._shadow_ws.force_split3()
                self._shadow_ws.submit_reads_arrays(batch.offset[sel], batch.length[sel], batch.seq, batch.fi, batch.ri, batch.fp, batch.rp,
                                                    batch.fn[sel], batch.rn[sel], site_counts=cnt[sel], seed=self.seed,
                                                    stream=self._slots[k].stream, read_keys=batch.name_hash[sel])
                shadowed = True
            self._inflight.append((k, job, sel, shadowed))
            job["pending"] += 1
            start += take
            self._turn += 1
        return job

    def probs_of_native_batch(self, batch, skip=None):
        """The raw (n_sites, 2) probabilities of one bamnative.Batch in the model's CURRENT arithmetic (synchronous; chunks as
        feed_native_batch cuts them): what the data probe compares between two arithmetics (DeviceModel.data_probe)."""
        cnt = np.where(batch.length > 0, batch.n_sites, 0).astype(np.int64)
        if skip is not None:
            cnt[np.asarray(skip, bool)] = 0
        idx = np.flatnonzero(cnt > 0)
        out, start = [], 0
        while start < len(idx):
            csum = np.cumsum(cnt[idx[start:]])
            take = max(1, int(np.searchsorted(csum, self.batch_size, side="right")))
            sel = idx[start:start + take]
            csites = int(cnt[sel].sum())
            if self._rwss[0] is None or self._rwss[0].max_sites < csites:
                if self._rwss[0] is not None:
                    self._rwss[0].close()
                self._rwss[0] = self.dm.workspace(max(csites, self.batch_size))
            self._rwss[0].submit_reads_arrays(batch.offset[sel], batch.length[sel], batch.seq, batch.fi, batch.ri, batch.fp, batch.rp,
                                              batch.fn[sel], batch.rn[sel], site_counts=cnt[sel], seed=self.seed,
                                              stream=self._slots[0].stream, read_keys=batch.name_hash[sel])
            out.append(self._rwss[0].wait_reads()[3].copy())
            start += take
        return np.concatenate(out) if out else np.empty((0, 2), np.float32)

    def _shadow_ws_free(self):
        return not any(sh for _, _, _, sh in self._inflight)      # one shadow chunk in flight at a time (one shadow workspace)

    def _collect_one(self):
        k, job, sel, shadowed = self._inflight.pop(0)
        f, lc, _, pr = self._rwss[k].wait_reads()
        if not np.array_equal(np.diff(f), job["cnt"][sel]):
            raise RuntimeError("device and host site counts disagree")
        if shadowed:
            ps = self._shadow_ws.wait_reads()[3]
            d = float(np.abs(np.asarray(pr, np.float32) - np.asarray(ps, np.float32)).max()) if len(pr) else 0.0
            self.shadow_chunks += 1
            self.shadow_sites += len(pr)
            self.shadow_max = max(self.shadow_max, d)
            if not (d <= self.shadow_limit):
                raise ArithmeticViolation("split-mx left the rule on this input: max |dprob| %.3g against split3 over a shadowed chunk of %d sites "
                                          "(limit %.3g; shadow chunk %d)" % (d, len(pr), self.shadow_limit, self.shadow_chunks))
        a = int(job["first"][sel[0]])
        job["locs"][a:a + len(lc)] = lc                 # sel is a run of consecutive usable reads: their spans are adjacent
        job["prob1"][a:a + len(lc)] = prob1_norm_round6(pr)
        job["pending"] -= 1

    def finish_native_batch(self, job):
        """-> (first_site int32 (n_reads+1), locs int32, prob1 float32, tagged uint8 (n_reads), n_failed) of a fed batch, in its read
        order: exactly the arguments of NativeBamWriter.write_batch.  Blocks only for chunks of this job that are still in flight."""
        while job["pending"]:
            self._collect_one()
        return job["first"], job["locs"], job["prob1"], job["tagged"], job["failed"]

    def run_native_batch(self, batch, skip=None):
        """feed + finish in one call (drains the GPU at the end of the batch)."""
        return self.finish_native_batch(self.feed_native_batch(batch, skip))

    # ---- host side ------------------------------------------------------------------------------------------------
    def run(self, reads):
        """reads: sequence of Read (e.g. one or many hole-batches).  Returns ([ReadCalls per read, input order], n_failed).
        Sites are packed into device batches of `batch_size` regardless of read boundaries; batch i+1 is extracted and
        staged while batch i runs (two workspaces, two streams)."""
        reads = list(reads)
        if self.extract == "device":
            return self._run_device(reads)
        acc = [[] for _ in reads]
        failed = 0
        keys = ("kmer1", "ipd1", "pw1", "npass1", "kmer2", "ipd2", "pw2", "npass2", "site_key", "site_sub")
        cur = {k: [] for k in keys}
        meta, nsites, turn = [], 0, 0

        def flush():
            nonlocal cur, meta, nsites, turn
            if nsites == 0:
                return
            feats = {k: np.concatenate(v) for k, v in cur.items()}
            slot = self._slots[turn & 1]
            self._collect(slot, acc)                 # the batch submitted two turns ago on this slot
            self._submit(slot, feats, meta)
            turn += 1
            cur = {k: [] for k in keys}
            meta, nsites = [], 0

        for ridx, read in enumerate(reads):
            arr = extract_read_arrays(read.seq, read.fi, read.ri, read.fp, read.rp, no_decode=self.no_decode, norm=self.norm)
            if arr is None or len(arr["loc"]) == 0:
                failed += 1                            # reference counts reads without features as failed (:416-429)
                continue
            k = len(arr["loc"])
            start = 0
            while start < k:                          # a read may straddle device batches
                take = min(k - start, self.batch_size - nsites)
                sl = slice(start, start + take)
                cur["kmer1"].append(arr["fkmer"][sl]); cur["ipd1"].append(arr["fipd"][sl]); cur["pw1"].append(arr["fpw"][sl])
                cur["npass1"].append(np.full(take, read.fn, np.float32))
                cur["kmer2"].append(arr["rkmer"][sl]); cur["ipd2"].append(arr["ripd"][sl]); cur["pw2"].append(arr["rpw"][sl])
                cur["npass2"].append(np.full(take, read.rn, np.float32))
                cur["site_key"].append(np.full(take, read_key(read.name), np.uint64))
                cur["site_sub"].append(np.asarray(arr["loc"][sl], np.uint32))
                meta.append((ridx, arr["loc"][sl]))
                nsites += take
                start += take
                if nsites == self.batch_size:
                    flush()
        flush()
        for slot in (self._slots[turn & 1], self._slots[(turn + 1) & 1]):
            self._collect(slot, acc)
        out = []
        for read, parts in zip(reads, acc):
            if parts:
                out.append(_tags_for_read(read, np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])))
            else:
                out.append(ReadCalls(read.name, 0, np.empty(0, np.int64), np.empty(0, np.float32), None, None, 0))
        return out, failed


class HostNullPipe:
    """CCSM_NULL_MODEL=2 (diagnostics, tools/host_feed_probe.py): CallModsPipeline's native-batch interface without any GPU work - the
    kept CG sites from a host scan of the forward sequences, every probability 0.5.  What is left of call_mods is exactly its host
    side (BGZF inflate, record parse, MM/ML encoding, BGZF deflate, hand-out, stitching, index), with no GPU shared between the
    ranks of a probe that runs them all on one device."""

    def run_native_batch(self, batch, skip=None):
        nr = batch.n_reads
        cnt = np.where(batch.length > 0, batch.n_sites, 0).astype(np.int64)
        if skip is not None:
            cnt[np.asarray(skip, bool)] = 0
        first = np.zeros(nr + 1, np.int32)
        np.cumsum(cnt, out=first[1:])
        seq = batch.seq
        hit = np.flatnonzero((seq[:-1] == 67) & (seq[1:] == 71)) if len(seq) > 1 else np.empty(0, np.int64)
        rid = np.searchsorted(batch.offset, hit, side="right") - 1          # reads are laid out back to back
        loc = hit - batch.offset[rid]
        n = batch.length[rid].astype(np.int64)
        keep = (cnt[rid] > 0) & (loc + 1 < n) & (loc >= 10) & (loc < n - 10) & (n - 2 - loc >= 10) & (n - 2 - loc < n - 10)
        locs = loc[keep].astype(np.int32)
        if len(locs) != int(first[-1]):
            raise RuntimeError("host scan and reader site counts disagree")
        return first, locs, np.full(len(locs), 0.5, np.float32), (cnt > 0).astype(np.uint8), int(nr - np.count_nonzero(cnt > 0))

    def close(self):
        pass


def _tags_for_read(read, locs, probs):
    """call_modifications.py:230-263: sort by loc, MM over the forward sequence, ML = floor(p*256); a failed assertion
    leaves the read untagged (mm_flag 0)."""
    order = np.argsort(locs, kind="stable")
    locs, probs = locs[order], probs[order]
    try:
        mm = _convert_locs_to_mmtag(locs.tolist(), read.seq)     # read.seq is already the forward sequence
        ml = _convert_probs_to_mltag(list(probs))
        return ReadCalls(read.name, len(locs), locs, probs, mm, ml, 1)
    except AssertionError:
        return ReadCalls(read.name, len(locs), locs, probs, None, None, 0)
