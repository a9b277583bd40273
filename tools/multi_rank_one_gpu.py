"""`call_mods` under torch.distributed.run with N ranks on ONE GPU (every rank on cuda:0): the multi-GPU code path - chunk board, the data
probe's verdict and the end-of-file offset from rank 0, stitched output, index - with the real model; header and records must equal the
single-process run's.
usage: python tools/multi_rank_one_gpu.py [ranks=8] [reads=6000]"""
import hashlib, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from collections import OrderedDict
from ccsmeth_amd.utils import benchdata, synth

ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
tmp = tempfile.mkdtemp(prefix="ccsm_mr_")
inp, ckpt = os.path.join(tmp, "in.bam"), os.path.join(tmp, "m.ckpt")
benchdata.write_synthetic_hifi_bam(inp, reads, 15000, seed=5)
torch.save(OrderedDict((k, torch.from_numpy(v)) for k, v in synth.synth_weights(5).items()), ckpt)
digest = {}
for world in (1, ranks):
    out = os.path.join(tmp, "out")          # the same path both times: the @PG line quotes the command line
    base = ["-m", "ccsmeth_amd", "call_mods", "-i", inp, "-m", ckpt, "-o", out, "--batch_size", "12288", "--chunk_mb", "8"]
    cmd = [sys.executable] + base if world == 1 else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                                                      "--master-addr", "127.0.0.1", "--master-port", "29731"] + base
    env = dict(os.environ, PYTHONPATH=ROOT, CCSM_DEVICE_OVERRIDE="0")
    t0 = time.time()
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    dt = time.time() - t0
    lines = [ln for ln in p.stderr.splitlines() if ln.startswith("[main]arithmetic") or "wrote" in ln or "costs" in ln]
    print("world %d: rc %d, %.1f s\n  " % (world, p.returncode, dt) + "\n  ".join(lines), flush=True)
    if p.returncode != 0:
        print(p.stderr[-3000:])
        sys.exit(1)
    # the inflated BAM stream (header + records): independent of where the BGZF blocks were cut (a rank's chunks are block-aligned runs)
    import gzip
    digest[world] = hashlib.sha256(gzip.decompress(open(out + ".modbam.bam", "rb").read())).hexdigest()
    assert os.path.getsize(out + ".modbam.bam.bai") > 8
print("header and records identical:", digest[1] == digest[ranks])
assert digest[1] == digest[ranks]
