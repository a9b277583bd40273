"""Split-mx self-test on the GPU box: one pair product end to end + host vs device fp6 encoders, then a small forward vs the oracle."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ccsmeth_amd import _lib
lib = _lib.load()
for fmt in (2, 4):
    a, b, m = C.c_float(), C.c_float(), C.c_int()
    _lib.check(lib.ccsm_selftest_split_mx(0, fmt, C.byref(a), C.byref(b), C.byref(m)))
    print("split-mx selftest (weight blob fmt %d): max err with corr %.3e, main only %.3e, blob mismatching bytes %d" % (fmt, a.value, b.value, m.value))
a2, b2 = C.c_float(), C.c_float()
_lib.check(lib.ccsm_selftest_split_f8(0, C.byref(a2), C.byref(b2)))
print("split-f8 selftest: max err with corr %.3e, main only %.3e" % (a2.value, b2.value))
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.utils import synth
from oracle import attbigru2s_oracle as orc
for n in (96, 777):
    w = synth.synth_weights(7); s = synth.synth_sites(n, 8); h1, h2 = synth.synth_h0(n, 9)
    ref = orc.attbigru2s_forward(w, s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h1, h2)[1]
    for prec in (4, 3):
        dm = DeviceModel(w, 0, precision=prec); ws = dm.workspace(n)
        p = ws.forward_host(s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h0=(h1, h2))[1]
        d = np.abs(p - ref)
        print("n %d precision %d: max |dprob| %.3e  mean %.3e  finite %s" % (n, prec, d.max(), d.mean(), np.isfinite(p).all()))
        ws.close(); dm.close()
