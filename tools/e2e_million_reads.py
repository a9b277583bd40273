"""BASELINE configs[2] at a size that means something: `call_mods` BAM -> modbam over >= 1 M synthetic 15-kb HiFi reads
(~6e8 CpG sites; the config names 10 M reads: scale-down factor printed) on ONE GPU with a committed TRAINED checkpoint (what a user's
checkpoint is served with), as a child process whose resident set size and whose device memory are sampled every few seconds - both have to
stay flat over the run - and whose output is counted at the end (records in = records out, every read tagged).
The input is NREADS_BASE generated reads, their records REP times over (benchdata.replicate_bam; read names repeat, which call_mods does not
mind); both are cut down if the scratch disk is too small, and the log says so.
usage: python tools/e2e_million_reads.py [--reads 1000000] [--ckpt planted7_5000] [--log gpurun_out/x.log]"""
import argparse, json, os, shutil, subprocess, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=1000000)
ap.add_argument("--base", type=int, default=16000)
ap.add_argument("--ckpt", default="planted7_5000", help="a fixture of tests/golden/trained, or 'synthetic' (random initialisation: split-mx)")
ap.add_argument("--log", default=None)
ap.add_argument("--tmp", default=os.environ.get("TMPDIR", "/tmp"))
ap.add_argument("--extra", default="", help="extra call_mods flags")
args = ap.parse_args()
log = open(args.log, "a") if args.log else None


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    if log:
        log.write(s + "\n"); log.flush()


import torch  # noqa: E402
from collections import OrderedDict  # noqa: E402
from ccsmeth_amd.utils import benchdata, synth  # noqa: E402
from ccsmeth_amd import bamnative as bn  # noqa: E402

tmp = tempfile.mkdtemp(prefix="ccsm_1m_", dir=args.tmp)
free = shutil.disk_usage(tmp).free
per_read = 56000                                                  # ~4.2 GiB per 80000 reads (profiles/r02_v_large_bam_end_to_end.log)
rep = max(1, -(-args.reads // args.base))
fit = int((free * 0.80) // (per_read * args.base * 1.12))         # input + 12 % output
if rep > fit:
    say("# scratch disk: %.0f GiB free: %d x %d reads do not fit, running %d x" % (free / 2 ** 30, rep, args.base, max(1, fit)))
    rep = max(1, fit)
n_reads = rep * args.base
say("# e2e_million_reads: %d reads = %d generated x %d; BASELINE configs[2] names 10 M reads: scaled down %.1f x; scratch %s (%.0f GiB free)" % (
    n_reads, args.base, rep, 1e7 / n_reads, tmp, free / 2 ** 30))
src, inp = os.path.join(tmp, "base.bam"), os.path.join(tmp, "in.bam")
t0 = time.time()
gen_s, _ = benchdata.write_synthetic_hifi_bam(src, args.base, 15000, planted=1.0)
size = benchdata.replicate_bam(src, inp, rep)
os.remove(src)
say("# input: %d reads, %.2f GiB of BGZF, generated in %.0f s + replicated in %.0f s" % (n_reads, size / 2 ** 30, gen_s, time.time() - t0 - gen_s))
ckpt = os.path.join(tmp, "m.ckpt")
wt = synth.synth_weights(5) if args.ckpt == "synthetic" else dict(np.load(os.path.join(ROOT, "tests", "golden", "trained", args.ckpt + ".npz")))
torch.save(OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in wt.items()), ckpt)
rep_json = os.path.join(tmp, "report.json")
out = os.path.join(tmp, "out")
cmd = [sys.executable, "-m", "ccsmeth_amd", "call_mods", "-i", inp, "-m", ckpt, "-o", out, "--batch_size", "12288", "--no_sort"] + args.extra.split()
env = dict(os.environ, PYTHONPATH=ROOT, CCSM_CALLMODS_REPORT=rep_json)
say("# " + " ".join(cmd))
t0 = time.time()
child = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
samples = []


def rss_of(pid):
    try:
        for line in open("/proc/%d/status" % pid):
            if line.startswith("VmRSS"):
                return int(line.split()[1]) / 1024.0
    except OSError:
        pass
    return float("nan")


def sample():
    while child.poll() is None:
        fr, tot = torch.cuda.mem_get_info(0)
        part = out + ".modbam.bam"
        samples.append((time.time() - t0, rss_of(child.pid), (tot - fr) / 2 ** 20, os.path.getsize(part) if os.path.exists(part) else 0))
        time.sleep(5.0)


torch.cuda.init()
base_used = (lambda fr_tot: (fr_tot[1] - fr_tot[0]) / 2 ** 20)(torch.cuda.mem_get_info(0))
th = threading.Thread(target=sample, daemon=True)
th.start()
err = child.stderr.read().decode()
rc = child.wait()
wall = time.time() - t0
th.join(timeout=10)
for line in err.splitlines():
    if line.startswith("[main]") or "wrote" in line or "WARNING" in line:
        say("  | " + line)
if rc != 0:
    say("call_mods failed (rc %d):" % rc, err[-3000:])
    sys.exit(1)
d = json.load(open(rep_json))
say("# report: reads %d, tagged %d, failed %d, sites %d, work phase %.1f s -> %.3f M sites/s work phase, %.3f M sites/s whole run (%.1f s wall incl. model set-up)" % (
    d["reads"], d["tagged"], d["failed"], d["sites"], d["seconds_work"], d["sites"] / d["seconds_work"] / 1e6, d["sites"] / wall / 1e6, wall))
say("# t (s) | host RSS of the call_mods process (MiB) | device memory in use (MiB; %.0f before the run) | output bytes written (GiB) | sites/s since the previous sample (from output growth)" % base_used)
prev = None
total_out = os.path.getsize(out + ".modbam.bam")
for k, (t, rss, used, ob) in enumerate(samples):
    rate = ""
    if prev is not None and ob > prev[1] and total_out:
        rate = "%.2f M" % ((ob - prev[1]) / total_out * d["sites"] / (t - prev[0]) / 1e6)
    if k % 4 == 0 or k == len(samples) - 1:
        say("  %6.0f | %8.0f | %8.0f | %6.2f | %s" % (t, rss, used, ob / 2 ** 30, rate))
    prev = (t, ob)
steady = [s for s in samples if s[0] > 0.25 * wall]
if steady:
    say("# steady state (last three quarters of the run): host RSS %.0f .. %.0f MiB, device memory %.0f .. %.0f MiB" % (
        min(s[1] for s in steady), max(s[1] for s in steady), min(s[2] for s in steady), max(s[2] for s in steady)))
# the output: every record of the input, every read with sites tagged
n = tagged = 0
with bn.NativeBamReader(out + ".modbam.bam", threads=8) as rd:
    while True:
        b = rd.next_batch(4096)
        if b is None:
            break
        n += b.n_reads
        b.close()
say("# output: %d records (input %d), %.2f GiB; reported tagged %d" % (n, n_reads, total_out / 2 ** 30, d["tagged"]))
assert n == n_reads == d["reads"], (n, n_reads, d["reads"])
shutil.rmtree(tmp, ignore_errors=True)
say("# ok")
