"""NumPy emulation of the SPLIT_F8 arithmetic (fp16 main product + fp8 / fp6 correction products, quantised recurrent state)
inside the float64 oracle's GRU: the experiment that sized the scheme before ccsm_gru_f8.hip was written.
Test infrastructure (uses oracle/); CPU only, ~3 min.  usage: python tests/diag/emulate_split_f8.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import attbigru2s_oracle as orc
from ccsmeth_amd.utils import synth
def hi16(a): return a.astype(np.float16).astype(np.float32)
def lo16(a): return (a - hi16(a)).astype(np.float16).astype(np.float32)
def qfmt(a, mbits, emin, vmax):
    """round to a minifloat: mbits mantissa bits, min normal exponent emin (value 2^emin), saturate at vmax"""
    a = a.astype(np.float64); s = np.sign(a); v = np.abs(a)
    e = np.floor(np.log2(np.maximum(v, 1e-300))); e = np.maximum(e, emin)
    step = 2.0 ** (e - mbits)
    return (s * np.minimum(np.round(v / step) * step, vmax)).astype(np.float32)
def fp8(a): return qfmt(a, 3, -6, 448.0)
def fp6(a): return qfmt(a, 3, 0, 7.5)
def pow2_scale(maxabs, target):  # power of two s with maxabs*s <= target
    return 2.0 ** np.floor(np.log2(target / np.maximum(maxabs, 1e-30)))
def corr(xh, xl, wh, wl, fmt):
    L = 2.0 ** 11
    if fmt == "fp8":      # per-tensor weight scale, fixed activation scales
        sw = pow2_scale(np.abs(wh).max(), 240.0)
        a1, b1 = fp8(wl * L * sw), fp8(xh * 64.0)
        a2, b2 = fp8(wh * sw), fp8(xl * L * 64.0)
        return (b1 @ a1.T + b2 @ a2.T) / (L * sw * 64.0)
    if fmt == "fp6":      # weights: per (row, 32-k block) scale; activations fixed scale
        K = wh.shape[1]; out = 0
        for c in range(0, K, 32):
            whb, wlb = wh[:, c:c+32], wl[:, c:c+32] * L
            s1 = pow2_scale(np.abs(wlb).max(1, keepdims=True), 7.5); s2 = pow2_scale(np.abs(whb).max(1, keepdims=True), 7.5)
            a1 = fp6(wlb * s1) / s1; a2 = fp6(whb * s2) / s2
            b1 = fp6(xh[:, c:c+32] * 8.0) / 8.0; b2 = fp6(xl[:, c:c+32] * L * 8.0) / 8.0
            out = out + (b1 @ a1.T + b2 @ a2.T) / L
        return out
    if fmt == "fp6t":     # per-tensor weight scale too
        s2 = pow2_scale(np.abs(wh).max(), 7.5)
        a1 = fp6(wl * L * s2); a2 = fp6(wh * s2); b1 = fp6(xh * 8.0); b2 = fp6(xl * L * 8.0)
        return (b1 @ a1.T + b2 @ a2.T) / (L * s2 * 8.0)
def mm(x, w, mode):
    xh, xl, wh, wl = hi16(x), lo16(x), hi16(w), lo16(w)
    if mode == "full": return xh @ wh.T + xh @ wl.T + xl @ wh.T
    return xh @ wh.T + corr(xh, xl, wh, wl, mode)
MODE = {"x": "full", "h": "full", "x0": "full"}
def gru_direction(x, h0, w_ih, w_hh, b_ih, b_hh, reverse):
    n_b, seq_len, kin = x.shape; hid = h0.shape[1]
    h = h0.astype(np.float32, copy=True); out = np.empty((n_b, seq_len, hid), np.float32)
    gi_all = mm(x.reshape(n_b * seq_len, -1).astype(np.float32), w_ih, MODE["x"] if kin > 16 else MODE["x0"]).reshape(n_b, seq_len, -1) + b_ih
    for t in (range(seq_len - 1, -1, -1) if reverse else range(seq_len)):
        gi = gi_all[:, t, :]; gh = mm(h, w_hh, MODE["h"]) + b_hh
        r = orc._sigmoid(gi[:, :hid] + gh[:, :hid]); z = orc._sigmoid(gi[:, hid:2*hid] + gh[:, hid:2*hid])
        n = np.tanh(gi[:, 2*hid:] + r * gh[:, 2*hid:]); h = (h - n) * z + n
        if MODE.get("hq"): h = hi16(h) + fp8(lo16(h) * 2.0**17) / 2.0**17
        out[:, t, :] = h
    return out, h
orig = orc.gru_direction
n = 512
for wseed in (7, 11, 5):
    w = synth.synth_weights(wseed); s = synth.synth_sites(n, 3 + wseed); h1, h2 = synth.synth_h0(n, 4)
    f = lambda: orc.attbigru2s_forward(w, s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h1, h2)[1]
    orc.gru_direction = orig; ref = f()
    orc.gru_direction = gru_direction
    for m, hq in (("full", 0), ("fp8", 0), ("fp8", 1), ("fp6", 0), ("fp6t", 0)):
        MODE["x"] = MODE["h"] = m; MODE["hq"] = hq
        print(wseed, m, hq, "%.2e" % np.abs(f() - ref).max(), flush=True)
