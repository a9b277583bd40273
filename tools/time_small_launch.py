"""Per-kernel times of launches of 1, 2, 3, 4, 5 and 6 coalesced 2048-site batches (and of a 512-site call) in the workgroup form
launch_run picks and in each forced form (CCSM_WG_TILES = 1 | 2 | 3 -> 32 / 64 / 96 rows per workgroup).
usage: python tools/time_small_launch.py [precision=4]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.utils import synth
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
dm = DeviceModel(synth.synth_weights(7), 0, precision=prec)
for n, g in ((512, 1), (2048, 1), (2048, 2), (2048, 3), (2048, 4), (2048, 5), (2048, 6)):
    s = synth.synth_sites(n * g, 8)
    batches = []
    for b in range(g):
        sl = slice(b * n, (b + 1) * n)
        batches.append(tuple(torch.from_numpy(np.ascontiguousarray(s[k][sl])).to(dev) for k in
                             ("kmer1", "ipd1", "pw1", "npass1", "kmer2", "ipd2", "pw2", "npass2")))
    for form in ("picked", "1", "2", "3"):
        if form == "picked":
            os.environ.pop("CCSM_WG_TILES", None)
        else:
            os.environ["CCSM_WG_TILES"] = form
        ws = dm.workspace(n * g)
        ws.set_timing(True)
        for rep in range(24):
            for b in range(g):
                ws.group_add_torch(*batches[b], seed=1, offset=rep * n * g + b * n)
            ws.group_run()
        torch.cuda.synchronize()
        ms, nr = ws.timing_mean()
        print("%d x %4d sites, %-10s: gru0 %.3f gru1 %.3f gru2 %.3f attn %.3f  sum %.3f ms" % (
            g, n, "picked" if form == "picked" else "%d rows" % (32 * int(form)), ms[0], ms[1], ms[2], ms[3], float(np.sum(ms[:4]))))
        ws.close()
os.environ.pop("CCSM_WG_TILES", None)
