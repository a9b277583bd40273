"""Host-side mirror of the reference model interface for the call_mods hot path.

`ModelAttRNN` keeps the constructor signature, `forward` argument order, `state_dict` key names, `.eval()`,
`.cuda(dev)` and `get_model_type()` of reference ccsmeth/models.py:17-150, so that
ccsmeth/call_modifications.py:315-369 (construct, load checkpoint, strip a DDP "module." prefix, eval, cuda) and
:201-208 (the 16-tensor call) work unchanged, but every FLOP runs in libccsm's HIP kernels on gfx950.
There is no CPU path: constructing the device model without the extension or without a GPU raises.
"""
import ctypes as C
from collections import OrderedDict

import numpy as np

from . import _lib
from .utils.synth import state_dict_shapes


def _f32c(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


class DeviceModel:
    """Owns one ccsm_model (weights packed as MFMA fragments in HBM) and a pool of workspaces."""

    def __init__(self, state_dict, device=0, precision=0, model_type="attbigru2s", seq_len=21, num_layers=3,
                 num_classes=2, hidden_size=256, is_npass=True, is_sn=False, is_map=False, is_stds=False):
        lib = _lib.load()
        self._lib = lib
        self._keep = {k: _f32c(v) for k, v in state_dict.items()}
        self.features = (bool(is_npass), bool(is_stds), bool(is_sn), bool(is_map))
        k0 = 8 + 2 + int(bool(is_npass)) + 2 * int(bool(is_stds)) + 4 * int(bool(is_sn)) + int(bool(is_map))      # models.py:39-47
        if self._keep["rnn.weight_ih_l0"].shape[1] != k0:
            raise ValueError("rnn.weight_ih_l0 has %d columns, the feature flags need %d" % (self._keep["rnn.weight_ih_l0"].shape[1], k0))
        w = _lib.Weights()
        ptr = lambda k: self._keep[k].ctypes.data  # noqa: E731
        w.embed_weight = ptr("embed.weight")
        for layer in range(_lib.LAYERS):
            for d, sfx in enumerate(("", "_reverse")):
                w.weight_ih[layer][d] = ptr(f"rnn.weight_ih_l{layer}{sfx}")
                w.weight_hh[layer][d] = ptr(f"rnn.weight_hh_l{layer}{sfx}")
                w.bias_ih[layer][d] = ptr(f"rnn.bias_ih_l{layer}{sfx}")
                w.bias_hh[layer][d] = ptr(f"rnn.bias_hh_l{layer}{sfx}")
        w.att_wa, w.att_ua, w.att_va = ptr("_att3.Wa.weight"), ptr("_att3.Ua.weight"), ptr("_att3.va.weight")
        w.fc1_weight, w.fc1_bias = ptr("fc1.weight"), ptr("fc1.bias")
        cfg = _lib.Config(seq_len, num_layers, num_classes, hidden_size, int(is_npass), int(is_sn), int(is_map),
                          int(is_stds), model_type.encode(), int(precision))
        handle = C.c_void_p()
        _lib.check(lib.ccsm_create(C.byref(cfg), C.byref(w), int(device), C.byref(handle)))
        self.handle = handle
        self.device = int(device)
        self.precision = lib.ccsm_model_precision(handle)      # the arithmetic in use (3 after a fallback from the default)
        self.probe_error = float(lib.ccsm_model_probe_error(handle))
        self.probe_error_hybrid = float(lib.ccsm_model_probe_error_hybrid(handle))     # -1: split-mx was accepted (or a precision forced)
        self.probe_tail = float(lib.ccsm_model_probe_tail(handle, 4))                  # fraction of the probe sites beyond 1e-5
        self.probe_tail_hybrid = float(lib.ccsm_model_probe_tail(handle, 5))
        self.probe_error_mxd = float(lib.ccsm_model_probe_error_of(handle, 6))         # split-mx-d (-1: not run)
        self.probe_tail_mxd = float(lib.ccsm_model_probe_tail(handle, 6))
        self.probe_q999 = float(lib.ccsm_model_probe_q999(handle))                     # 99.9th percentile of |dprob| split-mx vs split3 over the probe sites
        self.probe_sites = int(lib.ccsm_model_probe_sites(handle))                     # probe sites run (0: a precision was forced)
        self.quant_error = float(lib.ccsm_model_quant_error(handle))
        self.auto_precision = int(precision) == 0
        self.data_probe_error, self.data_probe_q999, self.data_probe_sites, self.data_probe_verdict = -1.0, -1.0, 0, -1
        self._workspaces = []

    # ---- the selection rule on the caller's own data (include/ccsm.h: ccsm_model_data_probe_*) -------------------------------------
    def set_precision(self, precision):
        """Switch between the arithmetics whose weight streams are resident: split3 (3) always; 4 / 5 / 6 if ccsm_create probed or was
        asked for it."""
        _lib.check(self._lib.ccsm_model_set_precision(self.handle, int(precision)))
        self.precision = int(precision)

    def data_probe_add(self, probs_candidate, probs_split3):
        a = np.ascontiguousarray(probs_candidate, np.float32)
        b = np.ascontiguousarray(probs_split3, np.float32)
        if a.shape != b.shape or a.ndim != 2 or a.shape[1] != 2:
            raise ValueError("two (n, 2) probability arrays")
        _lib.check(self._lib.ccsm_model_data_probe_add(self.handle, a.ctypes.data, b.ctypes.data, int(a.shape[0])))

    def data_probe_decide(self):
        """-> the precision in use after the rule has seen every site added (split3 if the candidate failed on them)."""
        lib = self._lib
        self.precision = int(lib.ccsm_model_data_probe_decide(self.handle))
        self.data_probe_error = float(lib.ccsm_model_data_probe_error(self.handle))
        self.data_probe_q999 = float(lib.ccsm_model_data_probe_q999(self.handle))
        self.data_probe_sites = int(lib.ccsm_model_data_probe_sites(self.handle))
        self.data_probe_verdict = int(lib.ccsm_model_data_probe_verdict(self.handle))
        return self.precision

    def data_probe(self, run, max_sites=65536):
        """Apply ccsm_create's rule to the caller's own sites.  `run()` yields (n, 2) probability arrays of the SAME sites every time it is
        iterated (a generator function over the first batches of the input); it is iterated once in the arithmetic in use and once in
        split3.  No-op (returns the precision in use) unless the default picked split-mx."""
        if self.precision == _lib.PRECISION_SPLIT3:
            return self.precision
        cand = self.precision
        got = []
        try:
            for arith in (cand, _lib.PRECISION_SPLIT3):
                self.set_precision(arith)
                part, n = [], 0
                for p in run():
                    part.append(np.array(p, np.float32, copy=True))
                    n += len(p)
                    if n >= max_sites:
                        break
                got.append(np.concatenate(part) if part else np.empty((0, 2), np.float32))
        finally:
            self.set_precision(cand)            # (also when the second pass raises: the model must not be left in split3 by accident)
        if got[0].shape != got[1].shape:
            raise RuntimeError("data probe: the two passes saw different sites")
        self.data_probe_add(got[0][:max_sites], got[1][:max_sites])
        return self.data_probe_decide()

    def workspace(self, max_sites):
        ws = Workspace(self, max_sites)
        self._workspaces.append(ws)
        return ws

    def close(self):
        for ws in self._workspaces:
            ws.close()
        self._workspaces = []
        if self.handle:
            self._lib.ccsm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Workspace:
    def __init__(self, model, max_sites):
        self.model = model
        self.max_sites = int(max_sites)
        handle = C.c_void_p()
        _lib.check(model._lib.ccsm_workspace_create(model.handle, self.max_sites, C.byref(handle)))
        self.handle = handle
        self._keep = None

    def close(self):
        if self.handle:
            self.model._lib.ccsm_workspace_destroy(self.handle)
            self.handle = None

    def force_split3(self):
        """The next forward / submit / group run on this workspace runs in split3 whatever the model's arithmetic (one-shot)."""
        _lib.check(self.model._lib.ccsm_workspace_force_split3(self.handle))

    def _batch(self, kmer1, ipd1, pw1, npass1, kmer2, ipd2, pw2, npass2, ptr_of, extra=None):
        """extra: for a model with is_stds / is_sn / is_map, one dict per strand with the keys "ipd_std", "pw_std" (N, 21), "sn" (N, 4),
        "map" (N, 21) that the flags ask for; npass may be None without is_npass."""
        has_npass, has_stds, has_sn, has_map = self.model.features
        want = [k for k, on in (("ipd_std", has_stds), ("pw_std", has_stds), ("sn", has_sn), ("map", has_map)) if on]
        if want and (extra is None or len(extra) != 2):
            raise ValueError("this model variant needs extra=(strand1, strand2) dicts with %s" % want)
        b = _lib.Batch()
        n = None
        keep = []
        kmer_f32 = None
        per_base = None
        for s, (kmer, ipd, pw, npass) in enumerate(((kmer1, ipd1, pw1, npass1), (kmer2, ipd2, pw2, npass2))):
            kmer, kf = ptr_of(kmer, kmer=True)
            ipd, _ = ptr_of(ipd)
            pw, _ = ptr_of(pw)
            npass, _ = ptr_of(npass if has_npass else np.zeros(kmer[1][0], np.float32))
            keep += [kmer, ipd, pw, npass]
            ns = kmer[1][0]
            if n is None:
                n, kmer_f32 = ns, kf
                per_base = len(npass[1]) == 2
            if ns != n or kf != kmer_f32 or (len(npass[1]) == 2) != per_base:
                raise ValueError("both strands must have the same batch size and layouts")
            for t, name in ((kmer, "kmer"), (ipd, "ipd"), (pw, "pw")):
                if tuple(t[1]) != (n, _lib.SEQ_LEN):
                    raise ValueError("%s must have shape (N, %d)" % (name, _lib.SEQ_LEN))
            if tuple(npass[1]) not in ((n,), (n, _lib.SEQ_LEN)):
                raise ValueError("npass must have shape (N,) or (N, 21)")
            b.strand[s].kmer, b.strand[s].ipd, b.strand[s].pw, b.strand[s].npass = kmer[0], ipd[0], pw[0], npass[0]
            for key in want:
                t, _ = ptr_of(extra[s][key])
                if tuple(t[1]) != ((n, 4) if key == "sn" else (n, _lib.SEQ_LEN)):
                    raise ValueError("%s must have shape %s" % (key, "(N, 4)" if key == "sn" else "(N, 21)"))
                keep.append(t)
                setattr(b.strand[s], key, t[0])
        b.kmer_is_f32 = int(kmer_f32)
        b.npass_per_base = int(per_base)
        return b, n, keep

    @staticmethod
    def _h0(h0, n, ptr_of, seed, offset):
        h = _lib.H0()
        keep = []
        if h0 is None:
            h.mode = _lib.H0_DEVICE_RNG
        elif isinstance(h0, str) and h0 == "zero":
            h.mode = _lib.H0_ZERO
        else:
            h.mode = _lib.H0_EXPLICIT
            for s in range(2):
                t, _ = ptr_of(h0[s])
                if tuple(t[1]) != (2 * _lib.LAYERS, n, _lib.HIDDEN):
                    raise ValueError("h0 tensors must have shape (6, N, 256)")
                keep.append(t)
                h.h0[s] = t[0]
        h.seed, h.offset = int(seed), int(offset)
        return h, keep

    def forward_host(self, kmer1, ipd1, pw1, npass1, kmer2, ipd2, pw2, npass2, h0=None, seed=0, offset=0, stream=None, extra=None):
        """NumPy in / NumPy out through the pinned staging ring (ccsm_forward_host)."""
        def ptr_of(a, kmer=False):
            a = np.asarray(a)
            if kmer and a.dtype == np.uint8:
                a = np.ascontiguousarray(a)
                return (a.ctypes.data, a.shape, a), False
            a = _f32c(a)
            return (a.ctypes.data, a.shape, a), True
        b, n, keep = self._batch(kmer1, ipd1, pw1, npass1, kmer2, ipd2, pw2, npass2, ptr_of, extra)
        h, keep2 = self._h0(h0, n, ptr_of, seed, offset)
        logits = np.empty((n, 2), np.float32)
        probs = np.empty((n, 2), np.float32)
        _lib.check(self.model._lib.ccsm_forward_host(self.model.handle, self.handle, n, C.byref(b), C.byref(h),
                                                     logits.ctypes.data, probs.ctypes.data, stream))
        return logits, probs

    def forward_reads(self, reads, h0=None, seed=0, offset=0, stream=None, read_keys=None):
        """Read-level call (ccsm_forward_reads_host): feature extraction and the model on the GPU.

        reads: list of (seq str|bytes, fi, ri, fp, rp uint8 arrays of len(seq), fn, rn).  read_keys (uint64 per read): device-drawn
        initial states keyed by (read key, position of the C) instead of offset + running site index.  Returns (first_site int32
        (n_reads+1), locs int32 (n_sites), logits, probs float32 (n_sites, 2))."""
        nr = len(reads)
        lens = np.array([len(r[0]) for r in reads], np.int32)
        offs = np.zeros(nr, np.int64)
        offs[1:] = np.cumsum(lens[:-1], dtype=np.int64)
        cat = []
        for k in range(5):
            parts = []
            for r in reads:
                a = r[k]
                if k == 0:
                    a = np.frombuffer(a.encode("ascii") if isinstance(a, str) else bytes(a), np.uint8)
                else:
                    a = np.asarray(a)
                    if a.shape != (len(r[0]),):
                        raise ValueError("kinetics arrays must match the sequence length")
                    a = a.astype(np.uint8, copy=False)
                parts.append(a)
            cat.append(np.ascontiguousarray(np.concatenate(parts)))
        fn = np.array([r[5] for r in reads], np.float32)
        rn = np.array([r[6] for r in reads], np.float32)
        rd = _lib.Reads()
        rd.n_reads = nr
        rd.offset, rd.length = offs.ctypes.data, lens.ctypes.data
        rd.seq, rd.fi, rd.ri, rd.fp, rd.rp = (a.ctypes.data for a in cat)
        rd.fn, rd.rn = fn.ctypes.data, rn.ctypes.data
        if read_keys is not None:
            read_keys = np.ascontiguousarray(read_keys, np.uint64)
            if len(read_keys) != nr:
                raise ValueError("read_keys must have one entry per read")
            rd.h0_key = read_keys.ctypes.data
        h = _lib.H0()
        keep = []
        if h0 is None:
            h.mode = _lib.H0_DEVICE_RNG
        elif isinstance(h0, str) and h0 == "zero":
            h.mode = _lib.H0_ZERO
        else:
            from .extract_features import count_kept_sites
            n_exp = sum(count_kept_sites(cat[0][o:o + l]) for o, l in zip(offs, lens))
            h.mode = _lib.H0_EXPLICIT
            for s in range(2):
                t = _f32c(h0[s])
                if t.shape != (2 * _lib.LAYERS, n_exp, _lib.HIDDEN):
                    raise ValueError("h0 tensors must have shape (6, n_sites=%d, 256)" % n_exp)
                keep.append(t)
                h.h0[s] = t.ctypes.data
        h.seed, h.offset = int(seed), int(offset)
        first = np.zeros(nr + 1, np.int32)
        locs = np.empty(self.max_sites, np.int32)
        logits = np.empty((self.max_sites, 2), np.float32)
        probs = np.empty((self.max_sites, 2), np.float32)
        n = C.c_int32(0)
        _lib.check(self.model._lib.ccsm_forward_reads_host(self.model.handle, self.handle, C.byref(rd), C.byref(h),
                                                           first.ctypes.data, locs.ctypes.data, logits.ctypes.data,
                                                           probs.ctypes.data, C.byref(n), stream))
        n = n.value
        return first, locs[:n].copy(), logits[:n].copy(), probs[:n].copy()

    def forward_reads_arrays(self, offset, length, seq, fi, ri, fp, rp, fn, rn, seed=0, offset_counter=0, h0=None, stream=None):
        """ccsm_forward_reads_host on caller-owned arrays (e.g. the views of a bamnative.Batch): offset int64 / length int32 /
        fn, rn float32 per read, seq / fi / ri / fp / rp uint8 concatenated.  No copies.  Returns (first_site, locs, logits, probs)."""
        offset = np.ascontiguousarray(offset, np.int64)
        length = np.ascontiguousarray(length, np.int32)
        fn, rn = np.ascontiguousarray(fn, np.float32), np.ascontiguousarray(rn, np.float32)
        nr = len(offset)
        if not (len(length) == len(fn) == len(rn) == nr) or nr == 0:
            raise ValueError("per-read arrays must have the same, non-zero length")
        arrs = [np.ascontiguousarray(a, np.uint8) for a in (seq, fi, ri, fp, rp)]
        need = int((offset + length).max())
        if any(len(a) < need for a in arrs):
            raise ValueError("byte arrays are shorter than offset + length")
        rd = _lib.Reads()
        rd.n_reads = nr
        rd.offset, rd.length = offset.ctypes.data, length.ctypes.data
        rd.seq, rd.fi, rd.ri, rd.fp, rd.rp = (a.ctypes.data for a in arrs)
        rd.fn, rd.rn = fn.ctypes.data, rn.ctypes.data
        h = _lib.H0()
        if h0 is None:
            h.mode = _lib.H0_DEVICE_RNG
        elif isinstance(h0, str) and h0 == "zero":
            h.mode = _lib.H0_ZERO
        else:
            raise ValueError("forward_reads_arrays takes h0=None (device RNG) or 'zero'")
        h.seed, h.offset = int(seed), int(offset_counter)
        first = np.zeros(nr + 1, np.int32)
        locs = np.empty(self.max_sites, np.int32)
        logits = np.empty((self.max_sites, 2), np.float32)
        probs = np.empty((self.max_sites, 2), np.float32)
        n = C.c_int32(0)
        _lib.check(self.model._lib.ccsm_forward_reads_host(self.model.handle, self.handle, C.byref(rd), C.byref(h),
                                                           first.ctypes.data, locs.ctypes.data, logits.ctypes.data,
                                                           probs.ctypes.data, C.byref(n), stream))
        n = n.value
        return first, locs[:n], logits[:n], probs[:n]

    def submit_reads_arrays(self, offset, length, seq, fi, ri, fp, rp, fn, rn, site_counts=None, seed=0, offset_counter=0,
                            h0=None, stream=None, read_keys=None):
        """ccsm_submit_reads_host: enqueue one chunk of reads (arrays as forward_reads_arrays); with site_counts (int32 per
        read, e.g. bamnative.Batch.n_sites) the call does not wait for the GPU.  read_keys (uint64 per read, e.g.
        bamnative.Batch.name_hash): initial states keyed by (read key, position of the C).  Collect with wait_reads()."""
        offset = np.ascontiguousarray(offset, np.int64)
        length = np.ascontiguousarray(length, np.int32)
        fn, rn = np.ascontiguousarray(fn, np.float32), np.ascontiguousarray(rn, np.float32)
        nr = len(offset)
        if not (len(length) == len(fn) == len(rn) == nr) or nr == 0:
            raise ValueError("per-read arrays must have the same, non-zero length")
        arrs = [np.ascontiguousarray(a, np.uint8) for a in (seq, fi, ri, fp, rp)]
        if any(len(a) < int((offset + length).max()) for a in arrs):
            raise ValueError("byte arrays are shorter than offset + length")
        cnt = None
        if site_counts is not None:
            cnt = np.ascontiguousarray(site_counts, np.int32)
            if len(cnt) != nr:
                raise ValueError("site_counts must have one entry per read")
        rd = _lib.Reads()
        rd.n_reads = nr
        rd.offset, rd.length = offset.ctypes.data, length.ctypes.data
        rd.seq, rd.fi, rd.ri, rd.fp, rd.rp = (a.ctypes.data for a in arrs)
        rd.fn, rd.rn = fn.ctypes.data, rn.ctypes.data
        if read_keys is not None:
            read_keys = np.ascontiguousarray(read_keys, np.uint64)
            if len(read_keys) != nr:
                raise ValueError("read_keys must have one entry per read")
            rd.h0_key = read_keys.ctypes.data
        h = _lib.H0()
        if h0 is None:
            h.mode = _lib.H0_DEVICE_RNG
        elif isinstance(h0, str) and h0 == "zero":
            h.mode = _lib.H0_ZERO
        else:
            raise ValueError("submit_reads_arrays takes h0=None (device RNG) or 'zero'")
        h.seed, h.offset = int(seed), int(offset_counter)
        _lib.check(self.model._lib.ccsm_submit_reads_host(self.model.handle, self.handle, C.byref(rd), cnt.ctypes.data if cnt is not None else None,
                                                          C.byref(h), stream))
        self._reads_pending = nr

    def wait_reads(self):
        """ccsm_wait_reads_host -> (first_site, locs, logits, probs) of the chunk submitted last."""
        nr = self._reads_pending
        first = np.zeros(nr + 1, np.int32)
        locs = np.empty(self.max_sites, np.int32)
        logits = np.empty((self.max_sites, 2), np.float32)
        probs = np.empty((self.max_sites, 2), np.float32)
        n = C.c_int32(0)
        _lib.check(self.model._lib.ccsm_wait_reads_host(self.handle, first.ctypes.data, locs.ctypes.data, logits.ctypes.data,
                                                        probs.ctypes.data, C.byref(n)))
        n = n.value
        return first, locs[:n], logits[:n], probs[:n]

    def forward_torch(self, kmer1, ipd1, pw1, npass1, kmer2, ipd2, pw2, npass2, h0=None, seed=0, offset=0, stream=None,
                      out=None, extra=None):
        """torch CUDA tensors in / out, asynchronous on `stream` (default: torch's current stream)."""
        import torch
        dev = torch.device("cuda", self.model.device)

        def ptr_of(a, kmer=False):
            if not isinstance(a, torch.Tensor):
                a = torch.as_tensor(a)
            if kmer and a.dtype == torch.uint8:
                a = a.to(dev).contiguous()
                return (a.data_ptr(), tuple(a.shape), a), False
            a = a.to(device=dev, dtype=torch.float32).contiguous()
            return (a.data_ptr(), tuple(a.shape), a), True
        b, n, keep = self._batch(kmer1, ipd1, pw1, npass1, kmer2, ipd2, pw2, npass2, ptr_of, extra)
        h, keep2 = self._h0(h0, n, ptr_of, seed, offset)
        if out is None:
            logits = torch.empty((n, 2), dtype=torch.float32, device=dev)
            probs = torch.empty((n, 2), dtype=torch.float32, device=dev)
        else:
            logits, probs = out
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(self.model._lib.ccsm_forward_device(self.model.handle, self.handle, n, C.byref(b), C.byref(h),
                                                       logits.data_ptr(), probs.data_ptr(), stream))
        self._keep = (keep, keep2)   # inputs must outlive the asynchronous kernels
        return logits, probs

    def group_add_torch(self, kmer1, ipd1, pw1, npass1, kmer2, ipd2, pw2, npass2, h0=None, seed=0, offset=0, stream=None,
                        out=None):
        """Bind one device-resident batch to the next free rows of this workspace (ccsm_group_add_device)."""
        import torch
        dev = torch.device("cuda", self.model.device)

        def ptr_of(a, kmer=False):
            if not isinstance(a, torch.Tensor):
                a = torch.as_tensor(a)
            if kmer and a.dtype == torch.uint8:
                a = a.to(dev).contiguous()
                return (a.data_ptr(), tuple(a.shape), a), False
            a = a.to(device=dev, dtype=torch.float32).contiguous()
            return (a.data_ptr(), tuple(a.shape), a), True
        b, n, keep = self._batch(kmer1, ipd1, pw1, npass1, kmer2, ipd2, pw2, npass2, ptr_of)
        h, keep2 = self._h0(h0, n, ptr_of, seed, offset)
        if out is None:
            out = (torch.empty((n, 2), dtype=torch.float32, device=dev), torch.empty((n, 2), dtype=torch.float32, device=dev))
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(self.model._lib.ccsm_group_add_device(self.model.handle, self.handle, n, C.byref(b), C.byref(h),
                                                         out[0].data_ptr(), out[1].data_ptr(), stream))
        if self._keep is None or not isinstance(self._keep, list):
            self._keep = []
        self._keep.append((keep, keep2, out))
        return out

    def group_run(self, stream=None):
        """Run the heavy kernels once over every batch added since the last run (ccsm_group_run)."""
        if stream is None:
            import torch
            stream = torch.cuda.current_stream(torch.device("cuda", self.model.device)).cuda_stream
        _lib.check(self.model._lib.ccsm_group_run(self.model.handle, self.handle, stream))
        self._keep = list(self._keep[-16:]) if isinstance(self._keep, list) else self._keep

    def set_timing(self, enable=True):
        _lib.check(self.model._lib.ccsm_workspace_set_timing(self.handle, int(enable)))

    def timing_mean(self):
        """(mean ms [gru0, gru1, gru2, attn_fc, finalize], runs averaged) since set_timing(True)."""
        out = (C.c_float * 5)()
        n = C.c_int(0)
        _lib.check(self.model._lib.ccsm_workspace_timing_mean(self.handle, out, C.byref(n)))
        return list(out), n.value

    def last_timing(self):
        out = (C.c_float * 5)()
        _lib.check(self.model._lib.ccsm_workspace_last_timing(self.handle, out))
        return list(out)


class ModelAttRNN:
    """Drop-in for reference ccsmeth.models.ModelAttRNN(model_type="attbigru2s") on the inference path.

    Same constructor (models.py:18-22), same forward signature (models.py:89-90) returning (logits, softmax(logits)),
    same state_dict keys.  Extra keyword-only knobs, absent from the reference: `h0` on forward (pin the initial
    states; the reference draws them from an unseeded torch.randn — models.py:77-87), `precision` and `seed`."""

    def __init__(self, seq_len=21, num_layers=3, num_classes=2, dropout_rate=0.5, hidden_size=256,
                 is_npass=True, is_sn=False, is_map=False, is_stds=False, model_type="attbigru2s", device=0,
                 *, precision=0, seed=1234, max_batch=4096):
        if model_type not in ("attbigru2s",):
            raise ValueError("--model_type not set right!")      # models.py:57
        self.model_type = model_type
        self.device = device
        self.seq_len, self.num_layers, self.num_classes, self.hidden_size = seq_len, num_layers, num_classes, hidden_size
        self.is_npass, self.is_sn, self.is_map, self.is_stds = is_npass, is_sn, is_map, is_stds
        self.dropout_rate = dropout_rate   # identity at inference (model.eval())
        self.precision, self.seed, self.max_batch = precision, seed, max_batch
        feas_ccs = 2 + int(bool(is_npass)) + 2 * int(bool(is_stds)) + 4 * int(bool(is_sn)) + int(bool(is_map))      # models.py:39-47
        self._shapes = state_dict_shapes(seq_len, num_layers, num_classes, hidden_size, feas_ccs=feas_ccs)
        self._state = OrderedDict((k, np.zeros(s, np.float32)) for k, s in self._shapes.items())
        self._dev = None
        self._ws = None
        self._calls = 0

    coalesces_calls = True      # call_modifications._call_mods2s may hand this model more sites per call than --batch_size (same results)

    def get_model_type(self):
        return self.model_type

    # ---- checkpoint contract (call_modifications.py:342-358) -------------------------------------------------
    def state_dict(self):
        return OrderedDict((k, v.copy()) for k, v in self._state.items())

    def load_state_dict(self, sd):
        """Strict, like torch: a key/shape mismatch raises RuntimeError, which is what makes the reference retry
        with the 7-character "module." prefix stripped (call_modifications.py:350-358)."""
        def to_np(v):
            return v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        missing = [k for k in self._shapes if k not in sd]
        unexpected = [k for k in sd if k not in self._shapes]
        if missing or unexpected:
            raise RuntimeError("Error(s) in loading state_dict for ModelAttRNN: missing %s unexpected %s" % (missing, unexpected))
        for k, shape in self._shapes.items():
            a = to_np(sd[k])
            if tuple(a.shape) != tuple(shape):
                raise RuntimeError("size mismatch for %s: %s vs %s" % (k, tuple(a.shape), tuple(shape)))
            self._state[k] = np.ascontiguousarray(a, dtype=np.float32)
        self._release()

    def eval(self):
        return self

    def cuda(self, device=None):
        if device is not None:
            self.device = device
        self._ensure()
        return self

    def _release(self):
        if self._dev is not None:
            self._dev.close()
        self._dev = self._ws = None

    def _ensure(self, n=1):
        if self._dev is None:
            self._dev = DeviceModel(self._state, self.device, self.precision, self.model_type, self.seq_len,
                                    self.num_layers, self.num_classes, self.hidden_size, self.is_npass, self.is_sn,
                                    self.is_map, self.is_stds)
        if self._ws is None or self._ws.max_sites < n:
            if self._ws is not None:
                self._ws.close()
            self._ws = self._dev.workspace(max(n, self.max_batch))

    def forward(self, kmer, kpass, ipd_means, ipd_stds, pw_means, pw_stds, sns, maps,
                kmer2, kpass2, ipd_means2, ipd_stds2, pw_means2, pw_stds2, sns2, maps2, *, h0=None):
        """models.py:89-150.  ipd_stds / pw_stds / sns / maps are read when the model was built with is_stds / is_sn / is_map and
        ignored otherwise, exactly as in the reference."""
        import torch
        n = int(kmer.shape[0])
        self._ensure(n)
        is_torch = isinstance(kmer, torch.Tensor)
        extra = None
        if self.is_stds or self.is_sn or self.is_map:
            extra = ({"ipd_std": ipd_stds, "pw_std": pw_stds, "sn": sns, "map": maps},
                     {"ipd_std": ipd_stds2, "pw_std": pw_stds2, "sn": sns2, "map": maps2})
        if is_torch:
            logits, probs = self._ws.forward_torch(kmer, ipd_means, pw_means, kpass, kmer2, ipd_means2, pw_means2, kpass2,
                                                   h0=h0, seed=self.seed, offset=self._calls, extra=extra)
            if not kmer.is_cuda:   # reference on a CPU-tensor call returns CPU tensors
                logits, probs = logits.cpu(), probs.cpu()
        else:
            logits, probs = self._ws.forward_host(kmer, ipd_means, pw_means, kpass, kmer2, ipd_means2, pw_means2, kpass2,
                                                  h0=h0, seed=self.seed, offset=self._calls, extra=extra)
        self._calls += n
        return logits, probs

    __call__ = forward
