"""ctypes loader of oracle/_build/liboracle.so (the plain-C restatement).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "liboracle.so")


class _W(C.Structure):
    _fields_ = [("embed", C.c_void_p), ("w_ih", (C.c_void_p * 2) * 3), ("w_hh", (C.c_void_p * 2) * 3),
                ("b_ih", (C.c_void_p * 2) * 3), ("b_hh", (C.c_void_p * 2) * 3), ("wa", C.c_void_p), ("ua", C.c_void_p),
                ("va", C.c_void_p), ("fcw", C.c_void_p), ("fcb", C.c_void_p)]


_lib = None
DESCRIPTION = "oracle/attbigru2s_oracle.c fp32, blocked packed-panel GEMM + vectorised exp, OpenMP over 48-site blocks"


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise ImportError("oracle/_build/liboracle.so missing: run `make -C oracle`")
        _lib = C.CDLL(_PATH)
        _lib.oracle_forward.restype = C.c_int
        _lib.oracle_isa_name.restype = C.c_char_p
    return _lib


def max_threads():
    return int(load().oracle_max_threads())


def usable_threads():
    """Threads worth starting on this host: the OpenMP default capped by the scheduler affinity and by the cgroup CPU quota
    (a container that sees 256 CPUs but may use 16 CPU-seconds per second runs 16 threads at full speed and 128 at a fraction)."""
    import math
    n = max_threads()
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def block_sites():
    """Sites per OpenMP work item: size timing samples as a multiple of block_sites() * threads."""
    return int(load().oracle_block_sites())


def isa_name():
    """The instruction-set clone the library picks on this host ("avx512", "avx2+fma", "generic"; ORACLE_ISA overrides)."""
    return load().oracle_isa_name().decode()


def forward(weights, kmer1, ipd1, pw1, npass1, kmer2, ipd2, pw2, npass2, h0_1, h0_2, threads=0):
    """Same contract as attbigru2s_oracle.attbigru2s_forward (fp32).  Returns (logits, probs)."""
    lib = load()
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
    keep = {k: f(v) for k, v in weights.items()}
    w = _W()
    p = lambda k: keep[k].ctypes.data  # noqa: E731
    w.embed = p("embed.weight")
    for l in range(3):
        for d, sfx in enumerate(("", "_reverse")):
            w.w_ih[l][d] = p(f"rnn.weight_ih_l{l}{sfx}")
            w.w_hh[l][d] = p(f"rnn.weight_hh_l{l}{sfx}")
            w.b_ih[l][d] = p(f"rnn.bias_ih_l{l}{sfx}")
            w.b_hh[l][d] = p(f"rnn.bias_hh_l{l}{sfx}")
    w.wa, w.ua, w.va = p("_att3.Wa.weight"), p("_att3.Ua.weight"), p("_att3.va.weight")
    w.fcw, w.fcb = p("fc1.weight"), p("fc1.bias")
    n = int(np.asarray(kmer1).shape[0])
    k1 = np.ascontiguousarray(kmer1, dtype=np.uint8)
    k2 = np.ascontiguousarray(kmer2, dtype=np.uint8)
    arrs = [f(ipd1), f(pw1), f(npass1), f(ipd2), f(pw2), f(npass2), f(h0_1), f(h0_2)]
    assert arrs[2].shape == (n,) and arrs[6].shape == (6, n, 256)
    logits = np.empty((n, 2), np.float32)
    probs = np.empty((n, 2), np.float32)
    vp = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    rc = lib.oracle_forward(C.byref(w), n, vp(k1), vp(arrs[0]), vp(arrs[1]), vp(arrs[2]), vp(k2), vp(arrs[3]), vp(arrs[4]),
                            vp(arrs[5]), vp(arrs[6]), vp(arrs[7]), vp(logits), vp(probs), int(threads))
    if rc != 0:
        raise MemoryError("oracle_forward failed")
    return logits, probs
