"""NumPy emulation of the split-operand arithmetic with narrower correction operands (fp6 e2m3 / fp4 e2m1 on either side of
v_mfma_scale_f32_32x32x64_f8f6f4), inside the float64 oracle's GRU: sizes the accuracy of a format before any kernel is written
(the measured energy of each format pair is in profiles/r02_c_power_attribution.md).  Weights: one E8M0 scale per (row, 32-k
block) as the instruction allows; activations: fixed power-of-two scales.  The recurrent state is carried as fp16 hi + fp8 lo.
Test infrastructure (uses oracle/); CPU only.   usage: python tests/diag/emulate_corr_formats.py [n_sites] [heavy]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import attbigru2s_oracle as orc
from ccsmeth_amd.utils import synth


def hi16(a): return a.astype(np.float16).astype(np.float32)
def lo32(a): return (a - hi16(a)).astype(np.float32)


def qfmt(a, mbits, emin, vmax):
    a = a.astype(np.float64); s = np.sign(a); v = np.abs(a)
    e = np.floor(np.log2(np.maximum(v, 1e-300))); e = np.maximum(e, emin)
    step = 2.0 ** (e - mbits)
    return (s * np.minimum(np.round(v / step) * step, vmax)).astype(np.float32)


FMT = {"fp8": (3, -6, 448.0, 240.0), "fp6": (3, 0, 7.5, 7.5), "fp4": (1, 0, 6.0, 6.0)}


def q(a, f): return qfmt(a, *FMT[f][:3])
def pow2_scale(maxabs, target): return 2.0 ** np.floor(np.log2(target / np.maximum(maxabs, 1e-30)))


def corr(xh, xl, wh, wl, fa, fb, act_dyn):
    L = 2.0 ** 11
    K = wh.shape[1]; out = 0
    ta, tb = FMT[fa][3], FMT[fb][3]
    for c in range(0, K, 32):
        whb, wlb = wh[:, c:c + 32], wl[:, c:c + 32] * L
        s1 = pow2_scale(np.abs(wlb).max(1, keepdims=True), ta); s2 = pow2_scale(np.abs(whb).max(1, keepdims=True), ta)
        a1 = q(wlb * s1, fa) / s1; a2 = q(whb * s2, fa) / s2
        xhb, xlb = xh[:, c:c + 32], xl[:, c:c + 32] * L
        if act_dyn:     # per (row, block) scale from the block's largest magnitude
            t1 = pow2_scale(np.abs(xhb).max(1, keepdims=True), tb); t2 = pow2_scale(np.abs(xlb).max(1, keepdims=True), tb)
        else:           # |x_hi| < 1, |x_lo 2^11| <= 0.5 for |x| < 1 (GRU outputs)
            t1 = pow2_scale(1.0, tb); t2 = pow2_scale(0.5, tb)
        b1 = q(xhb * t1, fb) / t1; b2 = q(xlb * t2, fb) / t2
        out = out + (b1 @ a1.T + b2 @ a2.T) / L
    return out


def mm(x, w, mode):
    if isinstance(mode, dict):      # input part with per-gate weight formats: rows [0, 2H) = r, z gates, rows [2H, 3H) = n gate
        hid = w.shape[0] // 3
        return np.concatenate([mm(x, w[:2 * hid], mode["rz"]), mm(x, w[2 * hid:], mode["n"])], axis=1)
    xh, xl, wh, wl = hi16(x), lo32(x), hi16(w), lo32(w)
    if mode == "full": return xh @ wh.T + xh @ wl.T + xl @ wh.T
    if mode == "fp16": return xh @ wh.T
    fa, fb, dyn = mode
    return xh @ wh.T + corr(xh, xl, wh, wl, fa, fb, dyn)


MODE = {"x": "full", "h": "full", "hq": 0}


def gru_direction(x, h0, w_ih, w_hh, b_ih, b_hh, reverse):
    n_b, seq_len, kin = x.shape; hid = h0.shape[1]
    h = h0.astype(np.float32, copy=True); out = np.empty((n_b, seq_len, hid), np.float32)
    gi_all = mm(x.reshape(n_b * seq_len, -1).astype(np.float32), w_ih, MODE["x"] if kin > 16 else "full").reshape(n_b, seq_len, -1) + b_ih
    for t in (range(seq_len - 1, -1, -1) if reverse else range(seq_len)):
        gi = gi_all[:, t, :]; gh = mm(h, w_hh, MODE["h"]) + b_hh
        r = orc._sigmoid(gi[:, :hid] + gh[:, :hid]); z = orc._sigmoid(gi[:, hid:2 * hid] + gh[:, hid:2 * hid])
        n = np.tanh(gi[:, 2 * hid:] + r * gh[:, 2 * hid:]); h = (h - n) * z + n
        if MODE["hq"]: h = hi16(h) + q(lo32(h) * 2.0 ** 17, "fp8") / 2.0 ** 17
        out[:, t, :] = h
    return out, h


def heavy_tailed(w, seed):
    """Student-t (3 dof) entries at the synthetic scale, a few x50 outliers per matrix, gate-saturating biases"""
    rng = np.random.default_rng(seed)
    o = {}
    for k, v in w.items():
        if k.startswith("rnn.weight"):
            t = rng.standard_t(3, size=v.shape).astype(np.float32) * (np.abs(v).mean())
            idx = rng.integers(0, t.size, size=8)
            t.reshape(-1)[idx] *= 50.0
            o[k] = t
        elif k.startswith("rnn.bias"):
            o[k] = (v + rng.choice([-6.0, 0.0, 0.0, 6.0], size=v.shape)).astype(np.float32)
        else:
            o[k] = v
    return o


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    heavy = len(sys.argv) > 2
    orig = orc.gru_direction
    modes = [("full", 0), ("fp16", 0), (("fp8", "fp8", 0), 1), (("fp6", "fp6", 0), 1), (("fp6", "fp4", 0), 1), (("fp6", "fp4", 1), 1),
             (("fp4", "fp6", 0), 1), (("fp4", "fp4", 0), 1), (("fp4", "fp4", 1), 1)]
    # the product's choice: recurrent part fp4 weights; input part fp4 for the r and z gates, fp6 for the n gate (fp4 THERE is what
    # biases every site the same way), fp6 activations everywhere
    PRODUCT = ({"rz": ("fp4", "fp6", 0), "n": ("fp6", "fp6", 0)}, ("fp4", "fp6", 0))
    XN_FP4 = ({"rz": ("fp6", "fp6", 0), "n": ("fp4", "fp6", 0)}, ("fp4", "fp6", 0))
    for wseed in (7, 11):
        w = synth.synth_weights(wseed)
        if heavy: w = heavy_tailed(w, wseed)
        s = synth.synth_sites(n, 3 + wseed); h1, h2 = synth.synth_h0(n, 4)
        f = lambda: orc.attbigru2s_forward(w, s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h1, h2)[1]
        orc.gru_direction = orig; ref = f()
        orc.gru_direction = gru_direction
        for m, hq in modes:
            MODE["x"] = MODE["h"] = m; MODE["hq"] = hq
            d = np.abs(f() - ref)
            print("wseed %d %s%s  A=%s B=%s act_scale=%s : max |dprob| %.2e  99.9%% %.2e  mean %.2e" % (
                wseed, "heavy " if heavy else "", "hq" if hq else "  ", m if isinstance(m, str) else m[0], "" if isinstance(m, str) else m[1],
                "" if isinstance(m, str) else ("dyn" if m[2] else "fixed"), d.max(), np.quantile(d, 0.999), d.mean()), flush=True)
        for name, (xm, hm) in (("product (x: r,z fp4 | n fp6; h fp4)", PRODUCT), ("x: r,z fp6 | n fp4; h fp4", XN_FP4)):
            MODE["x"], MODE["h"], MODE["hq"] = xm, hm, 1
            d = f() - ref
            print("wseed %d %shq  %s : max |dprob| %.2e  mean |d| %.2e  mean d %+.2e" % (wseed, "heavy " if heavy else "", name, np.abs(d).max(),
                                                                                      np.abs(d).mean(), d[:, 1].mean()), flush=True)
