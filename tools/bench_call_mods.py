"""End-to-end `call_mods` throughput on a synthetic HiFi BAM: native I/O (libccsm_bam) vs the pure-Python reader/writer.
env: NREADS (1500), READLEN (15000), CPG (0.012 = fraction of positions made CG)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from collections import OrderedDict
from ccsmeth_amd import bamio
from ccsmeth_amd.call_mods import build_parser, call_mods
from ccsmeth_amd.utils import synth

nr, L, cpg = int(os.environ.get("NREADS", "1500")), int(os.environ.get("READLEN", "15000")), float(os.environ.get("CPG", "0.012"))
tmp = os.environ.get("TMPDIR", "/tmp")
inp = os.path.join(tmp, "bench_in.bam")
rng = np.random.default_rng(3)
t0 = time.time()
with bamio.BamWriter(inp, "@HD\tVN:1.5\tSO:unknown\n", [], level=1) as w:
    for i in range(nr):
        seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=L, p=[0.3, 0.2, 0.2, 0.3])
        pos = rng.integers(0, L - 1, int(L * cpg))
        seq[pos], seq[pos + 1] = ord("C"), ord("G")
        kin = lambda: np.clip(rng.gamma(2.0, 20.0, size=L), 0, 255).astype(np.uint8)  # noqa: E731
        tags = [("fi", "BC", kin()), ("fp", "BC", kin()), ("ri", "BC", kin()), ("rp", "BC", kin()), ("fn", "C", 12), ("rn", "C", 13), ("np", "C", 25)]
        w.write(bamio.BamRecord("m/%d/ccs" % i, flag=4, ref_id=-1, seq=seq.tobytes().decode(), qual=np.full(L, 40, np.uint8), tags=tags))
print("input: %d reads x %d bases, %.1f MB BAM (written in %.1f s)" % (nr, L, os.path.getsize(inp) / 1e6, time.time() - t0))
ckpt = os.path.join(tmp, "bench.ckpt")
torch.save(OrderedDict((k, torch.from_numpy(v)) for k, v in synth.synth_weights(5).items()), ckpt)
for io, extra in (("native", []), ("native", []), ("python", ["--io", "python"])):
    args = build_parser().parse_args(["-i", inp, "-m", ckpt, "-o", os.path.join(tmp, "bench_out_" + io), "--batch_size", "12288",
                                      "--holes_batch", "64"] + extra)
    t0 = time.time()
    res = call_mods(args, log=open(os.devnull, "w"))
    dt = time.time() - t0
    sites = 0
    with bamio.BamReader(res["output"]) as rd:
        for r in rd:
            if r.has_tag("ML"):
                sites += len(r.get_tag("ML"))
    print("--io %-6s: %.2f s wall (incl. model load), %d reads, %d sites -> %.0f reads/s, %.0f sites/s" % (io, dt, res["reads"], sites, res["reads"] / dt, sites / dt))
