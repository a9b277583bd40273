"""attbigru2s training on MI355X — host side of libccsm_train (include/ccsm_train.h) and the `trainm` loop.

Mirrors reference ccsmeth/train_multigpu.py: one process per GPU (`python -m torch.distributed.run --nproc-per-node N -m
ccsmeth_amd trainm ...`), every rank holds the full parameters, runs forward + backward on its share of each epoch's samples
(DistributedSampler semantics), the gradients are averaged with ONE flat 12.2 MB all-reduce over RCCL (DDP's buckets,
train_multigpu.py:171-172), then clip_grad_norm_(0.5) + Adam run identically on every rank (:309-312).  Validation loss is
all-reduced as the reference does (:46-50, 379).  The model arithmetic is libccsm_train's HIP path; there is no CPU fallback.
"""
import ctypes as C
import os
import sys
import time

import numpy as np

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CCSM_TRAIN_LIB_PATH") or os.path.join(_HERE, "lib", "libccsm_train.so")   # the variable: A/B builds (tools/)
EXPORTS = ("ccsm_train_last_error", "ccsm_train_num_params", "ccsm_train_param_offsets", "ccsm_train_create", "ccsm_train_destroy",
           "ccsm_train_forward_backward", "ccsm_train_eval", "ccsm_train_step", "ccsm_train_grad_ptr", "ccsm_train_get_params",
           "ccsm_train_set_params", "ccsm_train_get_grads", "ccsm_train_fused_fallbacks", "ccsm_train_selftest_gemm")

PARAM_NAMES = ["embed.weight"] + [f"rnn.{k}_l{l}{sfx}" for l in range(3) for sfx in ("", "_reverse")
                                  for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")] + \
    ["_att3.Wa.weight", "_att3.Ua.weight", "_att3.va.weight", "fc1.weight", "fc1.bias"]
PARAM_SHAPES = {"embed.weight": (5, 8), "_att3.Wa.weight": (256, 512), "_att3.Ua.weight": (256, 512), "_att3.va.weight": (1, 256),
                "fc1.weight": (2, 1024), "fc1.bias": (2,)}
for _l in range(3):
    for _sfx in ("", "_reverse"):
        PARAM_SHAPES[f"rnn.weight_ih_l{_l}{_sfx}"] = (768, 11 if _l == 0 else 512)
        PARAM_SHAPES[f"rnn.weight_hh_l{_l}{_sfx}"] = (768, 256)
        PARAM_SHAPES[f"rnn.bias_ih_l{_l}{_sfx}"] = (768,)
        PARAM_SHAPES[f"rnn.bias_hh_l{_l}{_sfx}"] = (768,)

_tl = None


def load():
    global _tl
    if _tl is not None:
        return _tl
    if not os.path.exists(LIB_PATH):
        raise ImportError("libccsm_train.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    try:
        import torch  # noqa: F401  (one HIP runtime in the process)
    except Exception:  # pragma: no cover
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    lib.ccsm_train_last_error.restype = C.c_char_p
    lib.ccsm_train_num_params.restype = C.c_int64
    lib.ccsm_train_param_offsets.argtypes = [vp, ci]
    lib.ccsm_train_create.argtypes = [C.POINTER(_lib.Weights), ci, ci, vp, C.POINTER(vp)]
    lib.ccsm_train_destroy.argtypes = [vp]
    lib.ccsm_train_destroy.restype = None
    lib.ccsm_train_forward_backward.argtypes = [vp, ci, C.POINTER(_lib.Batch), vp, C.POINTER(_lib.H0), cf, cf, C.c_uint64, C.POINTER(cf), vp]
    lib.ccsm_train_eval.argtypes = [vp, ci, C.POINTER(_lib.Batch), vp, C.POINTER(_lib.H0), cf, C.POINTER(cf), vp]
    lib.ccsm_train_step.argtypes = [vp, cf, cf, cf, cf, cf, C.POINTER(cf)]
    lib.ccsm_train_grad_ptr.argtypes = [vp, C.POINTER(vp)]
    lib.ccsm_train_get_params.argtypes = [vp, vp]
    lib.ccsm_train_set_params.argtypes = [vp, vp]
    lib.ccsm_train_get_grads.argtypes = [vp, vp]
    lib.ccsm_train_fused_fallbacks.argtypes = [vp]
    lib.ccsm_train_fused_fallbacks.restype = C.c_long
    lib.ccsm_train_selftest_gemm.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp, C.c_int, vp, C.c_int, C.c_float, vp, C.c_int, C.c_int]
    _tl = lib
    return lib


def _check(status):
    if status != _lib.OK:
        raise _lib.CcsmError(status, load().ccsm_train_last_error().decode())


def param_offsets():
    off = (C.c_int64 * 31)()
    assert load().ccsm_train_param_offsets(off, 31) == 0
    return list(off)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Trainer:
    """One ccsm_trainer: parameters, Adam moments and activation buffers for batches of up to max_sites sites on `device`.
    grads_tensor: optional torch CUDA float32 tensor of num_params elements that receives the gradients (for all_reduce)."""

    def __init__(self, state_dict, device=0, max_sites=512, grads_tensor=None):
        self._lib = load()
        sd = {}
        for k, v in state_dict.items():
            v = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            sd[k[7:] if k.startswith("module.") else k] = _f32(v)
        missing = [k for k in PARAM_NAMES if k not in sd]
        if missing:
            raise KeyError("state_dict lacks %s" % missing)
        for k in PARAM_NAMES:
            if tuple(sd[k].shape) != PARAM_SHAPES[k]:
                raise ValueError("%s has shape %s, expected %s" % (k, sd[k].shape, PARAM_SHAPES[k]))
        w = _lib.Weights()
        ptr = lambda k: sd[k].ctypes.data  # noqa: E731
        w.embed_weight = ptr("embed.weight")
        for layer in range(3):
            for d, sfx in enumerate(("", "_reverse")):
                w.weight_ih[layer][d] = ptr(f"rnn.weight_ih_l{layer}{sfx}")
                w.weight_hh[layer][d] = ptr(f"rnn.weight_hh_l{layer}{sfx}")
                w.bias_ih[layer][d] = ptr(f"rnn.bias_ih_l{layer}{sfx}")
                w.bias_hh[layer][d] = ptr(f"rnn.bias_hh_l{layer}{sfx}")
        w.att_wa, w.att_ua, w.att_va = ptr("_att3.Wa.weight"), ptr("_att3.Ua.weight"), ptr("_att3.va.weight")
        w.fc1_weight, w.fc1_bias = ptr("fc1.weight"), ptr("fc1.bias")
        self.num_params = int(self._lib.ccsm_train_num_params())
        self.offsets = param_offsets()
        self._grads_tensor = grads_tensor
        gptr = None
        if grads_tensor is not None:
            if grads_tensor.numel() != self.num_params or not grads_tensor.is_cuda or str(grads_tensor.dtype) != "torch.float32":
                raise ValueError("grads_tensor must be a CUDA float32 tensor of %d elements" % self.num_params)
            gptr = grads_tensor.data_ptr()
        self.handle = C.c_void_p()
        _check(self._lib.ccsm_train_create(C.byref(w), int(device), int(max_sites), gptr, C.byref(self.handle)))
        self.device, self.max_sites = int(device), int(max_sites)

    def close(self):
        if self.handle:
            self._lib.ccsm_train_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _batch(sites):
        """sites: dict with kmer1/2 (N,21) uint8, ipd1/2, pw1/2 (N,21) float32, npass1/2 (N,) float32."""
        b = _lib.Batch()
        keep = []
        n = len(sites["kmer1"])
        for s in (0, 1):
            km = np.ascontiguousarray(sites["kmer%d" % (s + 1)], dtype=np.uint8)
            ipd, pw, npass = _f32(sites["ipd%d" % (s + 1)]), _f32(sites["pw%d" % (s + 1)]), _f32(sites["npass%d" % (s + 1)])
            if km.shape != (n, 21) or ipd.shape != (n, 21) or pw.shape != (n, 21) or npass.shape not in ((n,), (n, 21)):
                raise ValueError("feature arrays must be (N, 21) and npass (N,) or (N, 21)")
            keep += [km, ipd, pw, npass]
            b.strand[s].kmer, b.strand[s].ipd, b.strand[s].pw, b.strand[s].npass = (km.ctypes.data, ipd.ctypes.data, pw.ctypes.data,
                                                                                   npass.ctypes.data)
            b.npass_per_base = int(npass.ndim == 2)
        b.kmer_is_f32 = 0
        return b, keep, n

    @staticmethod
    def _h0(h0, n, seed=0, offset=0):
        h = _lib.H0()
        keep = []
        if h0 is None:
            h.mode = _lib.H0_DEVICE_RNG
            h.seed, h.offset = int(seed), int(offset)
        elif isinstance(h0, str) and h0 == "zero":
            h.mode = _lib.H0_ZERO
        else:
            a, b = _f32(h0[0]), _f32(h0[1])
            if a.shape != (6, n, 256) or b.shape != (6, n, 256):
                raise ValueError("explicit h0 tensors must be (6, N, 256)")
            keep = [a, b]
            h.mode = _lib.H0_EXPLICIT
            h.h0[0], h.h0[1] = a.ctypes.data, b.ctypes.data
        return h, keep

    def forward_backward(self, sites, labels, h0=None, pos_weight=1.0, dropout_rate=0.0, seed=0, step=0, want_logits=False, h0_offset=None):
        """-> (loss, logits or None); the flat gradient buffer then holds this batch's gradients.
        h0_offset: running SITE index of the batch's first site in the device-drawn initial states (the generator's counter advances
        by 3072 per site); default step * N, so that consecutive steps draw disjoint windows (the reference draws a fresh
        torch.randn(6, N, 256) per forward: models.py:77-87)."""
        b, keep, n = self._batch(sites)
        lab = np.ascontiguousarray(labels, dtype=np.int32)
        if lab.shape != (n,):
            raise ValueError("labels must be (N,)")
        h, keep2 = self._h0(h0, n, seed, int(step) * n if h0_offset is None else int(h0_offset))
        loss = C.c_float()
        logits = np.empty((n, 2), np.float32) if want_logits else None
        _check(self._lib.ccsm_train_forward_backward(self.handle, n, C.byref(b), lab.ctypes.data, C.byref(h), float(pos_weight),
                                                     float(dropout_rate), int(seed) * 1000003 + int(step), C.byref(loss),
                                                     logits.ctypes.data if want_logits else None))
        return float(loss.value), logits

    def evaluate(self, sites, labels=None, h0=None, pos_weight=1.0, seed=0, step=0, h0_offset=None):
        b, keep, n = self._batch(sites)
        lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.int32)
        h, keep2 = self._h0(h0, n, seed, int(step) * n if h0_offset is None else int(h0_offset))
        loss = C.c_float()
        logits = np.empty((n, 2), np.float32)
        _check(self._lib.ccsm_train_eval(self.handle, n, C.byref(b), None if lab is None else lab.ctypes.data, C.byref(h),
                                         float(pos_weight), C.byref(loss), logits.ctypes.data))
        return float(loss.value), logits

    def step(self, lr, betas=(0.9, 0.999), eps=1e-8, max_norm=0.5):
        """clip_grad_norm_(max_norm) + Adam; -> gradient norm before clipping."""
        norm = C.c_float()
        _check(self._lib.ccsm_train_step(self.handle, float(lr), float(betas[0]), float(betas[1]), float(eps), float(max_norm), C.byref(norm)))
        return float(norm.value)

    def _split(self, flat):
        return {k: flat[self.offsets[i]:self.offsets[i + 1]].reshape(PARAM_SHAPES[k]).copy() for i, k in enumerate(PARAM_NAMES)}

    def state_dict(self):
        flat = np.empty(self.num_params, np.float32)
        _check(self._lib.ccsm_train_get_params(self.handle, flat.ctypes.data))
        return self._split(flat)

    @property
    def fused_fallbacks(self):
        """backward passes repeated step by step because a gate gradient left the fused kernels' range (include/ccsm_train.h)"""
        return int(self._lib.ccsm_train_fused_fallbacks(self.handle))

    def grads(self):
        flat = np.empty(self.num_params, np.float32)
        _check(self._lib.ccsm_train_get_grads(self.handle, flat.ctypes.data))
        return self._split(flat)

    def load_state_dict(self, sd):
        flat = np.concatenate([_f32(sd[k]).ravel() for k in PARAM_NAMES])
        assert flat.size == self.num_params
        _check(self._lib.ccsm_train_set_params(self.handle, flat.ctypes.data))


def selftest_gemm(a, b, c=None, t_a=False, t_b=False, alpha=1.0, beta=0.0, grad_a=False, device=0):
    """C = alpha op(A) op(B) + beta C through the training library's matrix-product kernel (include/ccsm_train.h); a, b: 2-D float32 arrays
    as STORED (t_a: a is K x M; t_b: b is N x K), possibly non-contiguous views along the first axis (their row stride is passed)."""
    lib = load()
    a, b = np.asarray(a), np.asarray(b)
    assert a.dtype == np.float32 and b.dtype == np.float32 and a.strides[1] == 4 and b.strides[1] == 4
    M, K = (a.shape[1], a.shape[0]) if t_a else a.shape
    N = b.shape[0] if t_b else b.shape[1]
    assert (b.shape[1] if t_b else b.shape[0]) == K
    out = np.zeros((M, N), np.float32) if c is None else np.ascontiguousarray(c, np.float32).copy()
    _check(lib.ccsm_train_selftest_gemm(int(device), int(t_a), int(t_b), M, N, K, float(alpha), a.ctypes.data, a.strides[0] // 4, b.ctypes.data,
                                        b.strides[0] // 4, float(beta), out.ctypes.data, N, int(grad_a)))
    return out
