"""What does the HOST side of multi-GPU `call_mods` sustain?  ("the host feature path, not xGMI, is the
scaling limiter".)

Runs `python -m ccsmeth_amd call_mods` on a synthetic HiFi BAM with 1 / 2 / 4 / 8 ranks on this one-GPU box with the model left out:
  CCSM_NULL_MODEL=2 ("host"): no GPU work at all - BGZF inflate, record parse, site scan, MM/ML, BGZF deflate, hand-out, stitching and
                    indexing; the ceiling the host pipeline alone puts on an N-GPU node with this many host cores;
  CCSM_NULL_MODEL=1 ("gpu-shared"): transfers, on-device feature extraction and initial states, result copies as well, only the
                    BiGRU / attention launches left out - all ranks share the ONE GPU here, so its time slicing between N processes is
                    in the figure (it would not be on N GPUs).
The same run with the real model on one rank gives the 1-GPU end-to-end rate beside them.
    python tools/host_feed_probe.py            env: NREADS (16000 generated), TIMES (6: the records are appended that often), READ_LEN (15000),
                                                    WORLDS ("1,2,4,8"), CORES (cgroup quota or cpu count), MODES ("2,1"), REAL ("1")"""
import json
import os
import resource
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from collections import OrderedDict  # noqa: E402
from ccsmeth_amd.utils import benchdata, synth  # noqa: E402


def host_cores():
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, int(int(q) / int(p)))
    except (OSError, ValueError):
        pass
    return len(os.sched_getaffinity(0))


tmp = os.environ.get("TMPDIR", "/tmp")
nreads, read_len = int(os.environ.get("NREADS", "16000")), int(os.environ.get("READ_LEN", "15000"))
inp = os.environ.get("KEEP_INPUT") or os.path.join(tmp, "hf_in.bam")
ckpt = os.path.join(tmp, "hf.ckpt")
cores = int(os.environ.get("CORES", "0")) or host_cores()
times = int(os.environ.get("TIMES", "6"))
if not os.path.exists(inp):
    base = inp + ".base"
    secs, size = benchdata.write_synthetic_hifi_bam(base, nreads, read_len)
    t0 = time.time()
    size = benchdata.replicate_bam(base, inp, times, threads=min(cores, 16))
    os.remove(base)
    print("# input: %d reads x %d bases generated in %.0f s, their records %d times over = %d reads, %.2f GiB compressed (+%.0f s)"
          % (nreads, read_len, secs, times, nreads * times, size / 2 ** 30, time.time() - t0))
else:
    print("# input: %s, %.2f GiB compressed (reused)" % (inp, os.path.getsize(inp) / 2 ** 30))
torch.save(OrderedDict((k, torch.from_numpy(v)) for k, v in synth.synth_weights(5).items()), ckpt)
print("# host: %d CPUs visible, %d CPU-seconds per second granted (cgroup cpu.max); every rank on cuda:0" % (os.cpu_count(), cores))
port = 29700


def run(world, null_model, threads, extra=()):
    global port
    port += 1
    rep = os.path.join(tmp, "hf_report.json")
    out = os.path.join(tmp, "hf_out_%d" % world)
    if os.path.exists(rep):
        os.remove(rep)
    cmd = [sys.executable, "-m", "ccsmeth_amd", "call_mods", "-i", inp, "-m", ckpt, "--batch_size", "12288", "--holes_batch", "256",
           "--threads", str(threads), "-o", out] + list(extra)
    r0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    t0 = time.time()
    procs = []
    for r in range(world):
        env = dict(os.environ, PYTHONPATH=ROOT, CCSM_CALLMODS_REPORT=rep, CCSM_NULL_MODEL=str(null_model))
        if world > 1:
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
    errs = [p.communicate(timeout=3600)[1] for p in procs]
    rc = [p.returncode for p in procs]
    wall = time.time() - t0
    r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    cpu = (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
    if any(rc):
        print("world %d FAILED rc %s\n%s" % (world, rc, "\n".join(e[-1500:] for e in errs)))
        return None
    d = json.load(open(rep))
    work = max(d.get("rank_seconds_work", [d["seconds_work"]]))
    print("world %d %-10s threads/rank %2d | %8d sites | work %6.2f s = %6.2f M sites/s | stitch %5.2f s index %5.2f s (%.1f %% of the run) | "
          "whole run %6.2f s = %6.2f M sites/s | process wall %5.1f s, %6.1f CPU-s = %.2f CPU-s per M sites, %4.1f cores busy"
          % (world, {0: "real-model", 1: "gpu-shared", 2: "host"}[null_model], threads, d["sites"], work, d["sites"] / work / 1e6, d["seconds_stitch"], d["seconds_index"],
             100.0 * (d["seconds_stitch"] + d["seconds_index"]) / d["seconds"], d["seconds"], d["sites"] / d["seconds"] / 1e6, wall, cpu, cpu / (d["sites"] / 1e6),
             cpu / wall), flush=True)
    for f in (out + ".modbam.bam", out + ".modbam.bam.bai"):
        if os.path.exists(f):
            os.remove(f)
    return d


worlds = [int(w) for w in os.environ.get("WORLDS", "1,2,4,8").split(",")]
run(1, 2, min(cores, 16))                         # warm the page cache and the libraries
if os.environ.get("REAL", "1") == "1":
    run(1, 0, min(cores, 16))
for mode in [int(m) for m in os.environ.get("MODES", "2,1").split(",")]:
    for w in worlds:
        run(w, mode, max(2, cores // w))
if not os.environ.get("KEEP_INPUT"):
    os.remove(inp)
