"""Why did `call_mods --arithmetic auto` (data probe -> split3) and `--arithmetic split3` write different bytes?  Per-read ML arrays of
several runs side by side."""
import io, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from collections import OrderedDict
from ccsmeth_amd import bamio
from ccsmeth_amd.call_mods import build_parser, call_mods
from ccsmeth_amd.utils import benchdata, synth
tmp = tempfile.mkdtemp(prefix="ccsm_dpb_")
tr = dict(np.load(os.path.join(ROOT, "tests", "golden", "trained", "planted11_12000_nodrop.npz"))); init = synth.synth_weights(11)
w = {k: (init[k] + np.float32(0.08) * (tr[k] - init[k])).astype(np.float32) for k in init}
inp, ckpt = os.path.join(tmp, "in.bam"), os.path.join(tmp, "m.ckpt")
benchdata.write_synthetic_hifi_bam(inp, 160, 15000, seed=11, planted=0.0)
torch.save(OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in w.items()), ckpt)
res = {}
for tag, extra in (("auto", []), ("split3", ["--arithmetic", "split3"]), ("split3b", ["--arithmetic", "split3"]), ("auto2", []), ("mx", ["--no_data_probe"])):
    log = io.StringIO()
    r = call_mods(build_parser().parse_args(["-i", inp, "-m", ckpt, "-o", os.path.join(tmp, tag), "--batch_size", "12288", "--no_sort"] + extra), log=log)
    with bamio.BamReader(r["output"]) as rd:
        res[tag] = [(rec.query_name, np.asarray(rec.get_tag("ML")) if rec.has_tag("ML") else np.empty(0, np.uint8)) for rec in rd]
    print(tag, [ln for ln in log.getvalue().splitlines() if "arithmetic" in ln][-1][:110], "bytes", os.path.getsize(r["output"]))
for a, b in (("split3", "split3b"), ("auto", "auto2"), ("auto", "split3"), ("mx", "split3")):
    nd, first = 0, None
    for i, ((n1, m1), (n2, m2)) in enumerate(zip(res[a], res[b])):
        assert n1 == n2 and len(m1) == len(m2)
        d = np.flatnonzero(m1 != m2)
        if len(d):
            nd += len(d)
            if first is None:
                first = (i, n1, int(d[0]), int(m1[d[0]]), int(m2[d[0]]), len(d), len(m1))
    print("%s vs %s: %d differing ML values; first: %s" % (a, b, nd, first))
