"""Which part of split-mx loses accuracy on a TRAINED checkpoint, and what the cheapest repair of the recurrent part would be (NumPy
emulation inside the float64 oracle, like emulate_corr_formats.py).  Input: a trained state dict saved by
SAVE_TRAINED=<file.npz> python tests/diag/gpu_trained_weights_parity.py.   usage: python tests/diag/emulate_hybrid_variants.py <trained.npz> [n_sites]
Result on 1024 sites (profiles/r02_t_hybrid_variants_emulation.log): the fp4 W_lo of the RECURRENT part alone reproduces split-mx's tail
(1.0e-4 with an otherwise exact model); three fp16 passes there (the hybrid that ships) 1.8e-5; two fp16 passes + W_lo h_hi as an fp8 x fp8
MX product 1.9e-5 (built and measured: 1 % faster than the hybrid - tools/experiments/hybrid_fp8_group); an fp6 h_hi operand 4.6e-5."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "diag"))
import emulate_corr_formats as E
from oracle import attbigru2s_oracle as orc
from ccsmeth_amd.utils import synth
w = dict(np.load(sys.argv[1]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
s = synth.synth_sites(n, 143); h1, h2 = synth.synth_h0(n, 144)
f = lambda: orc.attbigru2s_forward(w, s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h1, h2)[1]
orig = orc.gru_direction; ref = f()
orc.gru_direction = E.gru_direction
_mm = E.mm
def corr_wlo_only(xh, wl, fa, fb):
    L = 2.0 ** 11; K = wl.shape[1]; out = 0
    ta, tb = E.FMT[fa][3], E.FMT[fb][3]
    for c in range(0, K, 32):
        wlb = wl[:, c:c + 32] * L
        s1 = E.pow2_scale(np.abs(wlb).max(1, keepdims=True), ta)
        a1 = E.q(wlb * s1, fa) / s1
        t1 = E.pow2_scale(1.0, tb)
        b1 = E.q(xh[:, c:c + 32] * t1, fb) / t1
        out = out + (b1 @ a1.T) / L
    return out
def mm(x, w_, mode):
    if isinstance(mode, tuple) and mode[0] == "h2":
        xh, xl, wh, wl = E.hi16(x), E.lo32(x), E.hi16(w_), E.lo32(w_)
        return xh @ wh.T + E.hi16(xl) @ wh.T + corr_wlo_only(xh, wl, mode[1], mode[2])
    return _mm(x, w_, mode)
E.mm = mm
XP = {"rz": ("fp4", "fp6", 0), "n": ("fp6", "fp6", 0)}
for name, xm, hm, hq in (("split-mx (product)", XP, ("fp4", "fp6", 0), 1), ("hybrid (h: 3 passes, exact state)", XP, "full", 0),
                         ("h2 fp6 (hi*hi + W_hi*h_lo fp16 + W_lo fp6 * h_hi fp6)", XP, ("h2", "fp6", "fp6"), 0),
                         ("h2 fp4", XP, ("h2", "fp4", "fp6"), 0), ("h2 fp8", XP, ("h2", "fp8", "fp8"), 0),
                         ("x full | h2 fp4", "full", ("h2", "fp4", "fp6"), 0), ("x full | h full", "full", "full", 0),
                         ("h2 A fp8 x B fp6", XP, ("h2", "fp8", "fp6"), 0), ("x full | h2 A fp8 x B fp6", "full", ("h2", "fp8", "fp6"), 0),
                         ("x full | h2 A fp8 x B fp8", "full", ("h2", "fp8", "fp8"), 0), ("x full | h2 A fp6 x B fp6", "full", ("h2", "fp6", "fp6"), 0)):
    E.MODE["x"], E.MODE["h"], E.MODE["hq"] = xm, hm, hq
    d = np.abs(f() - ref)[:, 1]
    print("%-60s max %.2e  99.9%% %.2e  99%% %.2e  mean %.2e  >1e-5: %d" % (name, d.max(), np.quantile(d, .999), np.quantile(d, .99), d.mean(), (d > 1e-5).sum()), flush=True)
