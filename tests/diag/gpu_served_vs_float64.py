"""What `precision 0` serves to the committed trained checkpoints (split3, the real kernels through the C-ABI) against FLOAT64 arithmetic over
2^20 sites each, beside a plain fp32 evaluation of the same model (the reference's own arithmetic class) against the same float64
forward: how far apart two fp32-class evaluations of such a model lie at single sites, at the scale where the 1e-4 bar is decided.
The float64 / fp32 forwards are tests/diag/emulate_int8_corr.py's torch restatement on the GPU (checked against the NumPy oracle).
usage: python tests/diag/gpu_served_vs_float64.py [--sites N] [--out log]"""
import argparse, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import emulate_int8_corr as emu
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.utils import synth

ap = argparse.ArgumentParser()
ap.add_argument("--sites", type=int, default=1 << 20)
ap.add_argument("--out", default=None)
args = ap.parse_args()
log = open(args.out, "a") if args.out else None


def say(*a):
    line = " ".join(str(x) for x in a)
    print(line, flush=True)
    if log:
        log.write(line + "\n"); log.flush()


dev = torch.device("cuda:0")
B = 8192
nblk = max(1, args.sites // B)
say("# gpu_served_vs_float64: %d sites per checkpoint (blocks of %d: plain synthetic sites and sites with the planted signal alternate; explicit N(0,1) initial states)" % (nblk * B, B))
for name in ("toy41_960", "planted7_5000", "planted11_12000_nodrop"):
    wt = dict(np.load(os.path.join(ROOT, "tests", "golden", "trained", name + ".npz")))
    W = emu.prepare(wt, dev)
    dm = DeviceModel(wt, device=0)
    ws = dm.workspace(B)
    d = {"served": [], "fp32": []}
    t0 = time.time()
    for b in range(nblk):
        s = synth.synth_sites(B, 70000 + b) if b % 2 == 0 else synth.synth_labeled_sites(B, 70000 + b)[0]
        h1, h2 = synth.synth_h0(B, 90000 + b)
        h0s = (torch.as_tensor(h1, device=dev), torch.as_tensor(h2, device=dev))
        ref = emu.forward(W, s, h0s, "f64", "f64", dev)[:, 1].cpu().numpy()
        f32 = emu.forward(W, s, h0s, "f32", "f32", dev)[:, 1].cpu().numpy()
        _, probs = ws.forward_host(s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h0=(h1, h2))
        d["served"].append(np.abs(probs[:, 1].astype(np.float64) - ref))
        d["fp32"].append(np.abs(f32 - ref))
    say("== %s: precision 0 -> %d (probe max %.2e); %d sites, %.0f s" % (name, dm.precision, dm.probe_error, nblk * B, time.time() - t0))
    for k, label in (("served", "the library (split3, three fp16 passes)"), ("fp32", "plain fp32 products (torch on the GPU) ")):
        e = np.concatenate(d[k])
        say("   %s against float64: max %.2e | >1e-5 %6d  >2.5e-5 %5d  >5e-5 %4d  >1e-4 %3d | 99.9%% %.2e 99.99%% %.2e mean %.2e" % (
            label, e.max(), (e > 1e-5).sum(), (e > 2.5e-5).sum(), (e > 5e-5).sum(), (e > 1e-4).sum(), np.quantile(e, 0.999), np.quantile(e, 0.9999), e.mean()))
    dm.close()
