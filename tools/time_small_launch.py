"""Per-kernel times of SMALL launches (1, 2 and 4 coalesced 2048-site batches) in the 64-row and the 96-row workgroup form
(CCSM_NO_64ROW=1 forces the latter).   usage: python tools/time_small_launch.py [precision=4]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.utils import synth
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n, dev = 2048, torch.device("cuda:0")
dm = DeviceModel(synth.synth_weights(7), 0, precision=prec)
for g in (1, 2, 4, 6):
    s = synth.synth_sites(n * g, 8)
    batches = []
    for b in range(g):
        sl = slice(b * n, (b + 1) * n)
        batches.append(tuple(torch.from_numpy(np.ascontiguousarray(s[k][sl])).to(dev) for k in
                             ("kmer1", "ipd1", "pw1", "npass1", "kmer2", "ipd2", "pw2", "npass2")))
    for form in ("64-row where it pays", "96-row"):
        if form == "96-row":
            os.environ["CCSM_NO_64ROW"] = "1"
        else:
            os.environ.pop("CCSM_NO_64ROW", None)
        ws = dm.workspace(n * g)
        ws.set_timing(True)
        for rep in range(24):
            for b in range(g):
                ws.group_add_torch(*batches[b], seed=1, offset=rep * n * g + b * n)
            ws.group_run()
        torch.cuda.synchronize()
        ms, nr = ws.timing_mean()
        print("%d batches, %-20s: gru0 %.3f gru1 %.3f gru2 %.3f attn %.3f  sum %.3f ms (n=%d)" % (g, form, ms[0], ms[1], ms[2], ms[3], float(np.sum(ms[:4])), nr))
        ws.close()
