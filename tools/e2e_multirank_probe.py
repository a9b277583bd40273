"""Multi-rank `call_mods` on ONE GPU (every rank on cuda:0, bookkeeping over gloo): wall time, per-rank batch counts and inflated bytes for
1 / 2 / 4 ranks under dynamic and static dispatch — the hand-out and stitching overhead, not a scaling number (the ranks share the GPU).
env: NREADS (3000)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from collections import OrderedDict
from ccsmeth_amd.utils import benchdata, synth

tmp = os.environ.get("TMPDIR", "/tmp")
inp, ckpt = os.path.join(tmp, "mr_in.bam"), os.path.join(tmp, "mr.ckpt")
print("input:", benchdata.write_synthetic_hifi_bam(inp, int(os.environ.get("NREADS", "3000")), 15000))
torch.save(OrderedDict((k, torch.from_numpy(v)) for k, v in synth.synth_weights(5).items()), ckpt)
base = [sys.executable, "-m", "ccsmeth_amd", "call_mods", "-i", inp, "-m", ckpt, "--batch_size", "12288", "--holes_batch", "128", "--no_sort"]
port = 29600
for world, dispatch in ((1, "dynamic"), (2, "dynamic"), (4, "dynamic"), (4, "static")):
    rep = os.path.join(tmp, "mr_report.json")
    out = os.path.join(tmp, "mr_out_%d_%s" % (world, dispatch))
    port += 1
    t0 = time.time()
    procs = []
    for r in range(world):
        env = dict(os.environ, PYTHONPATH=ROOT, CCSM_CALLMODS_REPORT=rep)
        if world > 1:
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen(base + ["-o", out, "--dispatch", dispatch], cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
    rc = [p.wait(timeout=900) for p in procs]
    wall = time.time() - t0
    d = json.load(open(rep)) if os.path.exists(rep) and not any(rc) else {}
    print("world %d %-7s rc %s process wall %.1f s | in-run %.2f s, %s sites -> %.2f M sites/s | batches per rank %s | inflated MB per rank %s scan %s"
          % (world, dispatch, rc, wall, d.get("seconds", 0), d.get("sites"), d.get("sites", 0) / max(d.get("seconds", 1), 1e-9) / 1e6,
             d.get("rank_batches"), [round(b / 1e6) for b in d.get("rank_inflated_bytes", [])] or round(d.get("inflated_bytes", 0) / 1e6),
             round(d.get("scan_inflated_bytes", 0) / 1e6)))
    if os.path.exists(rep):
        os.remove(rep)
sizes = {f: os.path.getsize(os.path.join(tmp, f)) for f in os.listdir(tmp) if f.startswith("mr_out_") and f.endswith(".bam")}
print("outputs:", sizes)
