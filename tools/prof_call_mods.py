import os, sys, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from collections import OrderedDict
from ccsmeth_amd.call_mods import build_parser, call_mods
tmp = os.environ.get("TMPDIR", "/tmp")
inp = os.path.join(tmp, "bench_in.bam"); ckpt = os.path.join(tmp, "bench.ckpt")
args = build_parser().parse_args(["-i", inp, "-m", ckpt, "-o", os.path.join(tmp, "prof_out"), "--batch_size", "12288", "--holes_batch", "64"])
call_mods(args, log=open(os.devnull, "w"))
pr = cProfile.Profile(); pr.enable()
t0 = time.time(); call_mods(args, log=open(os.devnull, "w")); print("wall", time.time() - t0)
pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(22)
