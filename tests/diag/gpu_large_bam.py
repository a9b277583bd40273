"""`call_mods` on an input BAM past 4 GiB (64-bit file offsets in the reader, the writer, the hand-out board and the stitching): one rank and
two ranks (both on cuda:0, bookkeeping over gloo) must write the same records.   usage: NREADS=80000 python tests/diag/gpu_large_bam.py"""
import hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from collections import OrderedDict
from ccsmeth_amd import bamio
from ccsmeth_amd.utils import benchdata, synth

tmp = os.environ.get("TMPDIR", "/tmp")
inp, ckpt = os.path.join(tmp, "big_in.bam"), os.path.join(tmp, "big.ckpt")
t0 = time.time()
print("input:", benchdata.write_synthetic_hifi_bam(inp, int(os.environ.get("NREADS", "80000")), 15000), "%.1f GiB" % (os.path.getsize(inp) / 2 ** 30), flush=True)
torch.save(OrderedDict((k, torch.from_numpy(v)) for k, v in synth.synth_weights(5).items()), ckpt)
base = [sys.executable, "-m", "ccsmeth_amd", "call_mods", "-i", inp, "-m", ckpt, "--batch_size", "12288", "--no_sort"]
digests = {}
for world in (1, 2):
    rep = os.path.join(tmp, "big_report.json")
    out = os.path.join(tmp, "big_out_%d" % world)
    procs = []
    t0 = time.time()
    for r in range(world):
        env = dict(os.environ, PYTHONPATH=ROOT, CCSM_CALLMODS_REPORT=rep)
        if world > 1:
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29655")
        procs.append(subprocess.Popen(base + ["-o", out], cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE))
    rc = [p.wait(timeout=3000) for p in procs]
    wall = time.time() - t0
    if any(rc):
        print("world", world, "rc", rc, procs[0].stderr.read().decode()[-2000:])
        sys.exit(1)
    d = json.load(open(rep))
    os.remove(rep)
    path = out + ".modbam.bam"
    h = hashlib.sha256()
    n = sites = 0
    with bamio.BamReader(path) as rd:
        for rec in rd:
            n += 1
            ml = rec.get_tag("ML") if rec.has_tag("ML") else b""
            sites += len(ml)
            h.update(rec.query_name.encode()); h.update(bytes(bytearray(ml)))
    digests[world] = (n, sites, h.hexdigest())
    print("world %d: wall %.1f s, report %s, output %.2f GiB, %d records, %d ML values, sha256 %s" % (
        world, wall, {k: d.get(k) for k in ("reads", "tagged", "sites", "seconds")}, os.path.getsize(path) / 2 ** 30, n, sites, h.hexdigest()[:16]), flush=True)
assert digests[1] == digests[2], digests
print("one rank and two ranks wrote the same records")
