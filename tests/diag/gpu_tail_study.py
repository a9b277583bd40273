"""Per-site error TAILS of every arithmetic on trained checkpoints, at the scale that decides what `precision 0` may serve:
trains checkpoints with libccsm_train (toy label of the early tests, and the planted-signal label of synth.synth_labeled_sites), saves them,
and runs S sites (default 2^20, device-drawn initial states, the same for every arithmetic) through split3 and through every faster
arithmetic; per (checkpoint, arithmetic): max |dprob| vs split3, sites beyond 1e-5 / 2.5e-5 / 5e-5 / 1e-4, the distribution of the maximum
over blocks of 8192 sites (what a fresh 8192-site test sees), and what ccsm_create's probe measured and selected.
usage: python tests/diag/gpu_tail_study.py [--sites N] [--out DIR] [--load a.npz b.npz ...] [--quick]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ccsmeth_amd.utils import synth

ap = argparse.ArgumentParser()
ap.add_argument("--sites", type=int, default=1 << 20)
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tail_study"))
ap.add_argument("--load", nargs="*", default=[])
ap.add_argument("--quick", action="store_true")
ap.add_argument("--save", type=int, default=4, help="how many of the trained checkpoints to keep as .npz")
ap.add_argument("--arith", default="4,6,5")
args = ap.parse_args()
os.makedirs(args.out, exist_ok=True)
log = open(os.path.join(args.out, "tail_study.log"), "a")


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    log.write(s + "\n"); log.flush()


sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_trained_fixtures as MTF


def train(*a):
    return MTF.train(*a, say=say)


plan = MTF.PLAN + [
        ("toy5_320", 5, 320, "toy", 1e-3, 0.5), ("planted13_2000", 13, 2000, "planted", 1e-3, 0.5), ("toy17_320", 17, 320, "toy", 1e-3, 0.5)]
if args.quick:
    plan = [("toy41_960", 41, 960, "toy", 1e-3, 0.5), ("planted7_600", 7, 600, "planted", 1e-3, 0.5)]
ckpts = []
for p in args.load:
    ckpts.append((os.path.splitext(os.path.basename(p))[0], dict(np.load(p))))
if not args.load:
    for i, (name, wseed, steps, kind, lr, dr) in enumerate(plan):
        wt = train(name, wseed, steps, kind, lr, dr)
        if i < args.save:
            np.savez(os.path.join(args.out, name + ".npz"), **wt)
        ckpts.append((name, wt))
    ckpts.append(("synthetic_init_7", synth.synth_weights(7)))

from ccsmeth_amd.models import DeviceModel
B = 8192
nblk = max(1, args.sites // B)
t0 = time.time()
blocks = []
for b in range(nblk):                                   # half plain synthetic sites, half with the planted signal
    blocks.append(synth.synth_sites(B, 70000 + b) if b % 2 == 0 else synth.synth_labeled_sites(B, 70000 + b)[0])
say("evaluation sites: %d blocks of %d (%.1f s to generate)" % (nblk, B, time.time() - t0))


def run(dm, ws):
    out = np.empty((nblk, B), np.float32)
    for b, s in enumerate(blocks):
        _, probs = ws.forward_host(s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h0=None, seed=777, offset=b * B)
        out[b] = probs[:, 1]
    return out


ariths = [int(a) for a in args.arith.split(",")]
names = {3: "split3", 4: "split-mx", 5: "hybrid", 6: "split-mx-d", 7: "arith7", 8: "arith8"}
for name, wt in ckpts:
    dm = DeviceModel(wt, device=0, precision=0)
    sel = dm.precision
    say("== %s: probe selects %d (%s) | probe max/tail: mx %.2e/%.4f mxd %.2e/%.4f hybrid %.2e/%.4f" % (
        name, sel, names.get(sel, "?"), dm.probe_error, dm.probe_tail, dm.probe_error_mxd, dm.probe_tail_mxd, dm.probe_error_hybrid, dm.probe_tail_hybrid))
    dm.close()
    dm = DeviceModel(wt, device=0, precision=3)
    ws = dm.workspace(B)
    ref = run(dm, ws)
    dm.close()
    say("   reference split3: mean prob %.3f, share > 0.5: %.3f" % (ref.mean(), (ref > 0.5).mean()))
    for a in ariths:
        try:
            dm = DeviceModel(wt, device=0, precision=a)
        except Exception as e:                                   # an arithmetic this build does not have
            say("   %-10s not available (%s)" % (names.get(a, a), str(e)[:60]))
            continue
        ws = dm.workspace(B)
        t1 = time.time()
        d = np.abs(run(dm, ws) - ref)
        dt = time.time() - t1
        dm.close()
        bm = d.max(1)
        rec = dict(ckpt=name, arith=a, sites=int(d.size), max=float(d.max()), n1e5=int((d > 1e-5).sum()), n25=int((d > 2.5e-5).sum()), n5e5=int((d > 5e-5).sum()),
                   n1e4=int((d > 1e-4).sum()), q999=float(np.quantile(d, 0.999)), q9999=float(np.quantile(d, 0.9999)), blockmax_median=float(np.median(bm)),
                   blockmax_q90=float(np.quantile(bm, 0.9)), blocks_over_25=int((bm > 2.5e-5).sum()), blocks_over_5e5=int((bm > 5e-5).sum()), blocks=int(nblk),
                   selected=int(sel), sites_per_s_host=float(d.size / dt))
        say("   %-10s max %.2e | >1e-5 %6d  >2.5e-5 %5d  >5e-5 %4d  >1e-4 %3d of %d | 99.9%% %.2e 99.99%% %.2e | 8192-block max: median %.2e, 90%% %.2e, blocks >2.5e-5: %d, >5e-5: %d of %d" % (
            names.get(a, a), rec["max"], rec["n1e5"], rec["n25"], rec["n5e5"], rec["n1e4"], d.size, rec["q999"], rec["q9999"], rec["blockmax_median"], rec["blockmax_q90"],
            rec["blocks_over_25"], rec["blocks_over_5e5"], nblk))
        log.write("JSON " + json.dumps(rec) + "\n"); log.flush()
