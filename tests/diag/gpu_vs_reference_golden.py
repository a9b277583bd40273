"""max |dprob| of the HIP path against the REFERENCE's own outputs (tests/golden/forward_golden.*), per case and arithmetic."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.utils import synth
G = os.path.join(ROOT, "tests", "golden")
fwd = np.load(os.path.join(G, "forward_golden.npz")); meta = json.load(open(os.path.join(G, "forward_golden.json")))
for name in ("b21_n64", "b21_n513", "b21_n2048"):
    m = meta[name]
    w = synth.synth_weights(m["weight_seed"]); s = synth.synth_sites(m["n"], m["site_seed"]); h1, h2 = synth.synth_h0(m["n"], m["h0_seed"])
    for prec in (4, 3):
        dm = DeviceModel(w, 0, precision=prec); ws = dm.workspace(m["n"])
        lg, pr = ws.forward_host(s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h0=(h1, h2))
        print("%-10s precision %d: max |dprob| %.3e  max |dlogit| %.3e  sites with |dprob| > 1e-5: %d" %
              (name, prec, np.abs(pr - fwd[name + "_probs"]).max(), np.abs(lg - fwd[name + "_logits"]).max(), int((np.abs(pr - fwd[name + "_probs"]).max(1) > 1e-5).sum())))
        dm.close()
