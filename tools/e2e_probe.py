"""End-to-end call_mods on a larger synthetic BAM (NREADS, default 6000 x 15 kb): wall, sites/s for two --holes_batch values and a cProfile of
the native loop (on the GPU box: 2.0 of 2.6 s are spent waiting for the GPU, 0.3 s in model set-up)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import torch
from collections import OrderedDict
from ccsmeth_amd.call_mods import build_parser, call_mods
from ccsmeth_amd.utils import benchdata, synth
tmp = "/tmp"
inp = os.path.join(tmp, "e2e_in.bam"); ckpt = os.path.join(tmp, "e2e.ckpt")
print("gen", benchdata.write_synthetic_hifi_bam(inp, int(os.environ.get("NREADS", "6000")), 15000))
torch.save(OrderedDict((k, torch.from_numpy(v)) for k, v in synth.synth_weights(5).items()), ckpt)
for hb in (64, 128, 512, 2048):
    args = build_parser().parse_args(["-i", inp, "-m", ckpt, "-o", os.path.join(tmp, "e2e_out"), "--batch_size", "12288", "--holes_batch", str(hb), "--no_sort"])
    call_mods(args, log=open(os.devnull, "w"))
    t0 = time.time(); res = call_mods(args, log=open(os.devnull, "w")); dt = time.time() - t0
    print("holes_batch", hb, "wall", dt, "sites", res.get("sites"), "sites/s", res.get("sites", 0) / dt, {k: v for k, v in res.items() if k not in ("output",)})
args = build_parser().parse_args(["-i", inp, "-m", ckpt, "-o", os.path.join(tmp, "e2e_out"), "--batch_size", "12288", "--holes_batch", "64", "--no_sort"])
pr = cProfile.Profile(); pr.enable()
call_mods(args, log=open(os.devnull, "w"))
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
