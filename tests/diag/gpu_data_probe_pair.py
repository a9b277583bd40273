"""Finds a (checkpoint, input) pair on which ccsm_create's SYNTHETIC probe accepts split-mx and the probe on the input's own sites
rejects it (the GPU test of VERDICT r04 item 3 pins one).  Checkpoints: W(a) = init + a (trained - init) between a committed trained
fixture and the initialisation it was trained from (tests/golden/make_trained_fixtures.py: planted7_5000 from synth_weights(7), toy41_960
from synth_weights(41), planted11_12000_nodrop from synth_weights(11)); input: a synthetic HiFi BAM with planted methylation kinetics.
usage: python tests/diag/gpu_data_probe_pair.py [--reads 160] [--planted 1.0]"""
import argparse, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ccsmeth_amd.utils import synth, benchdata  # noqa: E402
from ccsmeth_amd.models import DeviceModel  # noqa: E402
from ccsmeth_amd.pipeline import CallModsPipeline  # noqa: E402
from ccsmeth_amd.bamnative import NativeBamReader  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=160)
ap.add_argument("--planted", type=float, default=1.0)
ap.add_argument("--alphas", default="0.02,0.04,0.06,0.08,0.1,0.15,0.2,0.3")
args = ap.parse_args()
tmp = tempfile.mkdtemp(prefix="ccsm_dp_")
for planted in (0.0, args.planted):
    inp = os.path.join(tmp, "in%g.bam" % planted)
    benchdata.write_synthetic_hifi_bam(inp, args.reads, 15000, seed=11, planted=planted)
    for name, seed in (("planted7_5000", 7), ("toy41_960", 41), ("planted11_12000_nodrop", 11)):
        tr = dict(np.load(os.path.join(ROOT, "tests", "golden", "trained", name + ".npz")))
        init = synth.synth_weights(seed)
        for a in [float(x) for x in args.alphas.split(",")]:
            w = {k: (init[k] + a * (tr[k] - init[k])).astype(np.float32) for k in init}
            dm = DeviceModel(w, device=0)
            line = "%-24s planted %.1f alpha %.2f | synthetic probe: precision %d, max %.2e, 99.9%% %.2e over %d sites" % (
                name, planted, a, dm.precision, dm.probe_error, dm.probe_q999, dm.probe_sites)
            forced = False
            if dm.precision != 4:
                dm.set_precision(4); forced = True          # (what the data probe WOULD say)
            pipe = CallModsPipeline(dm, batch_size=12288, seed=1234, extract="device")
            with NativeBamReader(inp, threads=4) as rd:
                head, n = [], 0
                while n < 65536:
                    b = rd.next_batch(64)
                    if b is None:
                        break
                    head.append(b); n += int(np.where(b.length > 0, b.n_sites, 0).sum())
                dm.data_probe(lambda: (pipe.probs_of_native_batch(b) for b in head))
            print(line + " | data probe%s: max %.2e, 99.9%% %.2e over %d sites -> precision %d" % (
                " (forced)" if forced else "", dm.data_probe_error, dm.data_probe_q999, dm.data_probe_sites, dm.precision), flush=True)
            pipe.close(); dm.close()
