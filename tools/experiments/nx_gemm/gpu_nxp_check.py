import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.utils import synth
from oracle import attbigru2s_oracle as orc
w = synth.synth_weights(7)
n = 700
s = synth.synth_sites(n, 8); h1, h2 = synth.synth_h0(n, 9)
args = (s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"])
ref = orc.attbigru2s_forward(w, *args, h1, h2)[1]
dm = DeviceModel(w, device=0, precision=4)
p_nx = dm.workspace(n).forward_host(*args, h0=(h1, h2))[1]
os.environ["CCSM_NO_NXP"] = "1"
dm0 = DeviceModel(w, device=0, precision=4)
p_fused = dm0.workspace(n).forward_host(*args, h0=(h1, h2))[1]
del os.environ["CCSM_NO_NXP"]
print("nxp vs oracle %.3e | fused vs oracle %.3e | nxp vs fused %.3e" % (np.abs(p_nx - ref).max(), np.abs(p_fused - ref).max(), np.abs(p_nx - p_fused).max()))
for tiles in ("1", "2", "3"):
    os.environ["CCSM_WG_TILES"] = tiles
    p = dm.workspace(n).forward_host(*args, h0=(h1, h2))[1]
    print("form", tiles, "same bits as default:", np.array_equal(p, p_nx), "max diff %.2e" % np.abs(p - p_nx).max())
del os.environ["CCSM_WG_TILES"]
# timing: 6 x 2048 group through bench-like loop
import subprocess
