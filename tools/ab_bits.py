"""Do two builds of libccsm give the same BITS?  Run once per build (CCSM_LIB_PATH) with the same arguments; the second run compares with the first's file.
    CCSM_LIB_PATH=<a.so> python tools/ab_bits.py out.npy [precision] ; CCSM_LIB_PATH=<b.so> python tools/ab_bits.py out.npy [precision]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.utils import synth
path, prec = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4
out = []
dm = DeviceModel(synth.synth_weights(7), 0, precision=prec)
for n, form in ((6144, None), (1000, None), (513, "2"), (64, "1")):
    if form:
        os.environ["CCSM_WG_TILES"] = form
    else:
        os.environ.pop("CCSM_WG_TILES", None)
    s = synth.synth_sites(n, 500 + n)
    h1, h2 = synth.synth_h0(n, 501 + n)
    ws = dm.workspace(n)
    lg, pr = ws.forward_host(s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h0=(h1, h2))
    out.append(np.concatenate([np.asarray(lg).ravel(), np.asarray(pr).ravel()]))
    ws.close()
dm.close()
got = np.concatenate(out)
if os.path.exists(path):
    ref = np.load(path)
    same = ref.shape == got.shape and np.array_equal(ref.view(np.uint32), got.view(np.uint32))
    print("precision %d: %s (%d values; max |d| %.3g)" % (prec, "SAME BITS" if same else "DIFFERENT", got.size, float(np.abs(ref - got).max()) if ref.shape == got.shape else -1))
    sys.exit(0 if same else 1)
np.save(path, got)
print("precision %d: wrote %s (%d values)" % (prec, path, got.size))
