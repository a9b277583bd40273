"""Thread scaling of the CPU oracle on this host (sites/s and per thread) — sizing evidence for bench.py's cpu_baseline leg."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ccsmeth_amd.utils import synth  # noqa: E402
from oracle import c_oracle  # noqa: E402

print("isa", c_oracle.isa_name(), "block", c_oracle.block_sites(), "omp max threads", c_oracle.max_threads(), "os.cpu_count", os.cpu_count(),
      "affinity", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except OSError:
    pass
w = synth.synth_weights(1)
MAXT = c_oracle.max_threads()      # before the first call: oracle_forward(threads=k) sets the OpenMP default
for thr in [int(x) for x in (sys.argv[1:] or ["1", "8", "32", "64", "128"])]:
    if thr > MAXT:
        continue
    n = c_oracle.block_sites() * thr * 4
    s = synth.synth_sites(n, 2)
    h1, h2 = synth.synth_h0(n, 3)
    a = (s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h1, h2)
    c_oracle.forward(w, *a, threads=thr)
    t0 = time.perf_counter()
    c_oracle.forward(w, *a, threads=thr)
    dt = time.perf_counter() - t0
    print("%4d threads: %9.1f sites/s  %7.1f per thread  (%d sites, %.2f s)" % (thr, n / dt, n / dt / thr, n, dt), flush=True)
