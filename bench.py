#!/usr/bin/env python
"""bench.py — CpG sites/s of the MI355X-native attbigru2s call_mods hot path (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one forward of the hot path over one batch of 2048 synthetic CpG sites (both strands, 21-mers), features already
resident in HBM, initial states drawn on the device (Philox) as SURVEY.md 8(d) prescribes for the timed run.  `--coalesce`
steps are bound to one workspace and the heavy kernels run once over them (6 x 2048 sites = 512 workgroups of 96 strand rows
= two full rounds of the 256 CUs); a ragged last group (steps not a multiple of the group size) runs last, from its own workspace,
so that the per-kernel timers of the full groups describe full launches (running it on a second stream next to the last full
group was measured: no gain in wall time, inflated kernel durations).  Exactly K steps are timed between barrier + synchronize on both
sides, MAX over ranks is taken and rank 0 prints ONE JSON line.  Reads are sharded across GPUs with no collective on the data
path (weak scaling).  With --gpus N > 1 and no torch.distributed environment, this script re-executes itself under
torch.distributed.run with N ranks; it refuses to run if fewer than N GPUs are visible.

Besides BASELINE's metric the line carries (rank 0, N = 1; `--extras none` skips them): the same forward in the arithmetics
the rule did not select (split3: fp32-class, what a trained checkpoint gets - `extras.trained` runs a committed one), the PCIe-inclusive rate through ccsm_submit_host / ccsm_wait_host, `call_mods` end to end on a scaled-down
configs[2] BAM, and the aggregate kernel on configs[4]'s 50 M sites.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 2048
# Algorithmic work per CpG site: 2 strands x 21 steps x 2 directions x 768 gate rows x K columns MACs.
MAC_GRU0 = 2 * 21 * 2 * 768 * (11 + 256)
MAC_GRU12 = 2 * 21 * 2 * 768 * (512 + 256)          # per layer-1/2 launch: the dominant kernel
MAC_ATT = 2 * (21 * 512 * 256 + 512 * 256 + 21 * 256 + 21 * 512)
FLOP_PER_SITE = 2.0 * (MAC_GRU0 + 2 * MAC_GRU12 + MAC_ATT + 2048)   # = 244.23e6
BYTES_PER_SITE = 680.0
PEAK_F16_MFMA = 2.5e15       # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_HBM = 8.0e12
# HBM/fabric bytes of ONE launch of the dominant kernel per site: read from the file tools/pmc_summary.py --emit writes out of the round's
# rocprofv3 --pmc CSVs (separate FETCH_SIZE / WRITE_SIZE passes over the launch shape timed here - 6 x 2048 sites = 512 workgroups -, FETCH
# doubled per MI355X_MICROARCH.md's gfx950 note); PMC counters cannot be read from inside this process.  No file / no entry: traffic = null
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")


def traffic_of(precision, kernel):
    """-> (bytes per site of one launch of `kernel`, source) or (None, None)"""
    try:
        t = json.load(open(TRAFFIC_FILE))
    except (OSError, ValueError):
        return None, None
    for e in t.get("kernels", []):
        if e.get("precision") == precision and e.get("kernel") == kernel:
            return float(e["bytes_per_site"]), "%s (%s; %d sites per launch)" % (t.get("source", TRAFFIC_FILE), e.get("formula", ""), e.get("sites_per_launch", 0))
    return None, None


ARITH_NAME = {3: "split3", 4: "split-mx", 5: "hybrid", 6: "split-mx-d"}
ARITH = {4: ("fp32 reference; computed as f16 + MX(fp6|fp4 x fp6) split operands, f32 accumulate (within 1e-4 on this config's random-init weights)",
             "hi*hi on v_mfma_f32_32x32x16_f16 + (lo*hi, hi*lo) on v_mfma_scale_f32_32x32x64_f8f6f4 (GRU layers: weight blobs fp4 e2m1 for the "
             "recurrent part and the r, z gates' input part, fp6 e2m3 for the n gate's input part, per-(row, 32-k) E8M0 scales, fp6 e2m3 "
             "activation blobs; attention pool: fp8 e4m3 x fp8), one fp32 accumulator", 99.0 / 64.0),
         5: ("fp32 reference; computed as f16 + MX(fp6|fp4 x fp6) input part, f16x3 recurrent part, f32 accumulate",
             "GRU layers: the input part as in split-mx (hi*hi on v_mfma_f32_32x32x16_f16 + one block-scaled fp6/fp4 x fp6 correction MFMA per "
             "32 k), the recurrent part in three fp16 passes (hi*hi+hi*lo+lo*hi) on an fp16 hi + lo state; attention pool: fp8 e4m3 x fp8 "
             "correction; one fp32 accumulator", (512 * 99.0 / 64.0 + 256 * 3.0) / 768.0),
         6: ("fp32 reference; computed as f16 + MX(fp6 x fp6, per-row scales) split operands, f32 accumulate",
             "split-mx with fp6 e2m3 weight blobs for the recurrent part as well, and the state's fp6 correction blob scaled per (row, 32-k block) "
             "from the block's own largest magnitude (E8M0 from the fp16 hi fragments) instead of one fixed exponent", 99.0 / 64.0),
         3: ("fp32 reference; computed as f16x3 split operands (hi + lo), f32 accumulate: fp32-class", "split-fp16 x3 MFMA (hi*hi+hi*lo+lo*hi), fp32 accumulate", 3.0)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--coalesce", type=int, default=6,
                    help="batches run per launch of the heavy kernels (micro-batching).  6 x 2048 sites = 24576 strand rows = 512 GRU\n"
                         "workgroups (2 full rounds of the 256 CUs) and 768 attention workgroups (3 full rounds)")
    ap.add_argument("--precision", type=int, default=0, choices=(0, 3, 4, 5, 6),
                    help="0 = the library's default: split-mx (fp16 main product + MX correction product) if ccsm_create's 65536-site probe of these\n"
                         "weights against split3 is clean (max <= 1.25e-5, light tail: true of the contract's random initialisation), else split3\n"
                         "(three fp16 passes, fp32-class: what trained checkpoints get); 4 = split-mx forced; 6 = split-mx-d forced; 5 = hybrid forced; 3 = split3")
    ap.add_argument("--weights", default=None,
                    help="an .npz of state_dict arrays (e.g. SAVE_TRAINED=<file> python tests/diag/gpu_trained_weights_parity.py) instead of\n"
                         "the contract's random initialisation: what the probe selects for THAT checkpoint, and its speed")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--ceiling-seconds", type=float, default=3.0,
                    help="seconds per mode of the live power-capped MFMA ceiling probe behind roofline.peak_power_capped (0 = skip)")
    ap.add_argument("--extras", default="all", choices=("all", "none"), help="the secondary measurements (rank 0 at N = 1)")
    return ap.parse_args()


def respawn_under_torchrun(a):
    """--gpus N > 1 outside torch.distributed: check the devices, then run this script as N ranks."""
    import torch
    have = torch.cuda.device_count()
    if have < a.gpus:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible; refusing to report a %d-GPU number from fewer devices\n"
                         % (a.gpus, have, a.gpus))
        sys.exit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def cpu_baseline(weights, target_s, device_model):
    """The one leg that touches oracle/: oracle/attbigru2s_oracle.c on the host cores over a bounded sample of the same
    synthetic workload, plus (same leg, same probe inputs, h0 pinned) the max |delta prob| of the HIP path against it —
    the "prob delta vs ref" half of BASELINE.json's metric."""
    from ccsmeth_amd.utils import synth
    from oracle import c_oracle
    omp_default = c_oracle.max_threads()
    threads = c_oracle.usable_threads()          # OpenMP default capped by affinity and the cgroup CPU quota
    unit = c_oracle.block_sites() * threads          # one block per thread
    probe_n = unit
    s = synth.synth_sites(probe_n, 777)
    h1, h2 = synth.synth_h0(probe_n, 778)
    args = (s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h1, h2)
    c_oracle.forward(weights, *args, threads=threads)            # warm-up (thread pool, page faults)
    t0 = time.perf_counter()
    _, ref_probs = c_oracle.forward(weights, *args, threads=threads)
    rate = probe_n / (time.perf_counter() - t0)
    ws = device_model.workspace(probe_n)
    _, gpu_probs = ws.forward_host(s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"],
                                   h0=(h1, h2))
    ws.close()
    prob_err = float(np.abs(gpu_probs - ref_probs).max())
    n = int(min(max(rate * target_s, probe_n), 49152))      # bounded: the explicit h0 of 49152 sites is already 0.6 GB
    n = (n // unit) * unit or probe_n
    reps = max(1, int(round(rate * target_s / n)))
    s = synth.synth_sites(n, 779)
    h1, h2 = synth.synth_h0(n, 780)
    t0 = time.perf_counter()
    for _ in range(reps):
        _, ref_probs = c_oracle.forward(weights, s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h1, h2,
                                        threads=threads)
    dt = time.perf_counter() - t0
    n_total = n * reps
    # the HIP path against the oracle on the WHOLE timed sample (h0 pinned: 12 KiB per site from host memory, outside every timed region)
    ws = device_model.workspace(n)
    _, gpu_probs = ws.forward_host(s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h0=(h1, h2))
    ws.close()
    prob_err = max(prob_err, float(np.abs(gpu_probs - ref_probs).max()))
    return {"value": n_total / dt, "unit": "sites/s", "cores": threads, "kind": "port",
            "sites_per_s_per_core": n_total / dt / threads, "GFLOPs": n_total / dt * FLOP_PER_SITE / 1e9,
            "sample": "%d x %d synthetic sites (same generator as the GPU run), explicit h0, %s [%s], %d threads (OpenMP default %d, capped by "
                      "affinity / cgroup CPU quota), %.1f s" % (reps, n, c_oracle.DESCRIPTION, c_oracle.isa_name(), threads, omp_default, dt),
            "gpu_prob_max_abs_err": prob_err, "gpu_prob_err_sites": probe_n + n}


class Runner:
    """K steps of one DeviceModel.  Two workspaces on two streams take the groups in turn: the small per-batch kernels of a group (initial
    states, input packing: ccsm_group_add) run on their stream while the previous group's four heavy kernels run on the other; the
    heavy kernels themselves are chained by events, so exactly one of them runs at a time (their HIP-event times stay per-launch times)."""

    def __init__(self, dm, pool, dev, grp, rank):
        import torch
        self.torch, self.dm, self.pool, self.dev, self.grp, self.rank = torch, dm, pool, dev, grp, rank
        self.ws = [dm.workspace(BATCH * grp) for _ in range(3)]      # 0, 1: full groups in turn; 2: a ragged last group (own timers)
        self.overlap = os.environ.get("CCSM_BENCH_OVERLAP", "1") != "0"
        s0 = torch.cuda.Stream(dev)
        self.streams = [s0, torch.cuda.Stream(dev) if self.overlap else s0]
        self.outs = [[(torch.empty((BATCH, 2), device=dev), torch.empty((BATCH, 2), device=dev)) for _ in range(grp)] for _ in range(3)]
        self.step0 = 0
        self.k = 0
        self.last_run = None

    def run(self, steps):
        """Enqueue `steps` steps (no synchronisation)."""
        full, rag = divmod(steps, self.grp)
        i = self.step0
        for nb in [self.grp] * full + ([rag] if rag else []):
            st = self.streams[self.k]
            k = self.k if nb == self.grp else 2
            for j in range(nb):
                self.ws[k].group_add_torch(*self.pool[i % len(self.pool)], stream=st.cuda_stream, out=self.outs[k][j], seed=1234,
                                           offset=(self.rank * 10**9 + i * BATCH))
                i += 1
            if self.overlap and self.last_run is not None:
                st.wait_event(self.last_run)
            self.ws[k].group_run(stream=st.cuda_stream)
            if self.overlap:
                self.last_run = self.torch.cuda.Event()
                self.last_run.record(st)
            self.k ^= 1
        self.step0 = i
        return full, rag

    def arm(self):
        for w in self.ws:
            w.set_timing(True)

    def kernel_times(self, full_groups=True):
        """Mean per-launch kernel times (ms) of the full groups (both workspaces) or of the ragged group, and the launches averaged."""
        parts = [w.timing_mean() for w in (self.ws[:2] if full_groups else self.ws[2:])]
        n = sum(nr for _, nr in parts)
        if n == 0:
            return None, 0
        return sum(np.array(kt) * nr for kt, nr in parts if nr) / n, n

    def close(self):
        for w in self.ws:
            w.close()


MIN_WARM_GROUPS = int(os.environ.get("CCSM_BENCH_MIN_WARM_GROUPS", "10"))


def timed(runner, steps, warmup, fence):
    grp = runner.grp
    # warm-up: at least W steps and at least MIN_WARM_GROUPS full groups (the chip reaches its power-capped steady state after a few
    # tens of milliseconds: a timed region that starts on a cool, boosting chip does not describe sustained throughput), plus one group
    # of the ragged shape if the timed region has one
    w_steps = max(-(-warmup // grp), MIN_WARM_GROUPS) * grp + steps % grp
    runner.run(w_steps)
    fence()
    runner.arm()                 # average only the timed region's launches
    t0 = time.perf_counter()
    full, rag = runner.run(steps)
    fence()
    return time.perf_counter() - t0, full, rag, w_steps


def extras(weights, dm, dev, pool, grp):
    """Secondary measurements, each bounded to a few seconds; failures are reported, not raised."""
    import torch
    out = {}

    def leg(name, fn):
        try:
            out[name] = fn()
        except Exception as e:      # noqa: BLE001
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}

    def fence():
        torch.cuda.synchronize(dev)

    def other_arithmetic(prec):
        def run():
            from ccsmeth_amd.models import DeviceModel
            dmo = DeviceModel(weights, device=dev.index, precision=prec)
            r = Runner(dmo, pool, dev, grp, 0)
            steps = 4 * grp
            dt, full, _, _ = timed(r, steps, grp, fence)
            kt, nr = r.kernel_times()
            r.close(); dmo.close()
            ach = 2.0 * MAC_GRU12 * BATCH * grp / (float(np.mean(kt[1:3])) * 1e-3)
            return {"value": steps * BATCH / dt, "unit": "sites/s", "dtype": ARITH[prec][0], "steps": steps,
                    "roofline_frac": ach / PEAK_F16_MFMA, "launch_ms": float(np.mean(kt[1:3])), "mfma_passes_per_flop": ARITH[prec][2]}
        return run

    def trained():
        # What a user's TRAINED checkpoint gets: the committed checkpoint tests/golden/trained/planted7_5000.npz (5000 steps of libccsm_train
        # on the planted-signal label; fixed weights, so this leg is the same every run), served through ccsm_create(precision 0): the
        # arithmetic the rule selects for it (the three-pass one: on trained weights split-mx's probe is not clean), its rate on the
        # benchmark's workload, and its probabilities against the C oracle on 8192 sites (h0 pinned).
        from ccsmeth_amd.models import DeviceModel
        from ccsmeth_amd.utils import synth
        from oracle import c_oracle
        path = os.path.join(ROOT, "tests", "golden", "trained", "planted7_5000.npz")
        wt = dict(np.load(path))
        t0 = time.perf_counter()
        dmt = DeviceModel(wt, device=dev.index, precision=0)
        t_create = time.perf_counter() - t0
        r = Runner(dmt, pool, dev, grp, 0)
        steps = 4 * grp
        dt, _, _, _ = timed(r, steps, grp, fence)
        kt, _ = r.kernel_times()
        r.close()
        m = 8192
        sv = synth.synth_labeled_sites(m, 143)[0]
        h1, h2 = synth.synth_h0(m, 144)
        ws = dmt.workspace(m)
        _, gpu = ws.forward_host(sv["kmer1"], sv["ipd1"], sv["pw1"], sv["npass1"], sv["kmer2"], sv["ipd2"], sv["pw2"], sv["npass2"], h0=(h1, h2))
        ws.close()
        _, ref = c_oracle.forward(wt, sv["kmer1"], sv["ipd1"], sv["pw1"], sv["npass1"], sv["kmer2"], sv["ipd2"], sv["pw2"], sv["npass2"], h1, h2,
                                  threads=c_oracle.usable_threads())
        d = np.abs(gpu - ref)[:, 1]
        res = {"value": steps * BATCH / dt, "unit": "sites/s", "arithmetic_selected": ARITH_NAME.get(dmt.precision, dmt.precision),
               "probe": {"sites_run": dmt.probe_sites, "split_mx_max": dmt.probe_error, "split_mx_q999": dmt.probe_q999, "split_mx_tail_gt_1e-5": dmt.probe_tail,
                         "rule": "split-mx iff max <= 1.25e-5 and max <= 3 x q99.9 over 65536 probe sites, else split3"},
               "launch_ms": float(np.mean(kt[1:3])), "roofline_frac": 2.0 * MAC_GRU12 * BATCH * grp / (float(np.mean(kt[1:3])) * 1e-3) / PEAK_F16_MFMA,
               "max_abs_dprob_vs_oracle": float(d.max()), "sites_checked": m, "sites_beyond_1e-5": int((d > 1e-5).sum()), "sites_beyond_5e-5": int((d > 5e-5).sum()),
               "frac_called_methylated": float((ref[:, 1] > 0.5).mean()), "checkpoint": "tests/golden/trained/planted7_5000.npz", "create_seconds": t_create,
               "what": "a committed checkpoint trained by libccsm_train (recipe: tests/golden/make_trained_fixtures.py), served through ccsm_create(precision 0): "
                       "the arithmetic the rule selects, its rate on the benchmark's workload, max |dprob| against oracle/attbigru2s_oracle.c over 8192 sites (explicit h0)"}
        dmt.close()
        return res

    def pcie():
        # features in host memory every step, logits/probs back to host memory: ccsm_submit_host / ccsm_wait_host on two workspaces
        import ctypes as C
        from ccsmeth_amd import _lib
        from ccsmeth_amd.utils import synth
        n = BATCH * grp
        s = synth.synth_sites(n, 4242)
        lib = dm._lib
        wss = [dm.workspace(n) for _ in range(2)]
        sts = [torch.cuda.Stream(dev) for _ in range(2)]
        b = _lib.Batch()
        keep = []
        for k, sfx in enumerate(("1", "2")):
            arrs = [np.ascontiguousarray(s["kmer" + sfx], np.uint8), np.ascontiguousarray(s["ipd" + sfx], np.float32),
                    np.ascontiguousarray(s["pw" + sfx], np.float32), np.ascontiguousarray(s["npass" + sfx], np.float32)]
            keep += arrs
            b.strand[k].kmer, b.strand[k].ipd, b.strand[k].pw, b.strand[k].npass = (a.ctypes.data for a in arrs)
        b.kmer_is_f32, b.npass_per_base = 0, 0
        logits, probs = np.empty((n, 2), np.float32), np.empty((n, 2), np.float32)

        def submit(k, i):
            h = _lib.H0()
            h.mode, h.seed, h.offset = _lib.H0_DEVICE_RNG, 1234, i * n
            _lib.check(lib.ccsm_submit_host(dm.handle, wss[k].handle, n, C.byref(b), C.byref(h), sts[k].cuda_stream))

        def wait(k):
            _lib.check(lib.ccsm_wait_host(wss[k].handle, logits.ctypes.data, probs.ctypes.data))
        reps = 10
        for phase in (2, reps):
            fence()
            t0 = time.perf_counter()
            submit(0, 0)
            for i in range(1, phase):
                submit(i & 1, i)
                wait((i - 1) & 1)
            wait((phase - 1) & 1)
            dt = time.perf_counter() - t0
        for w in wss:
            w.close()
        return {"value": reps * n / dt, "unit": "sites/s", "sites_per_call": n, "calls": reps,
                "what": "ccsm_submit_host / ccsm_wait_host double-buffered: H2D of the 680 B/site features from host memory, model, D2H of logits + probs"}

    def call_mods_e2e():
        from ccsmeth_amd.utils import benchdata
        return benchdata.call_mods_end_to_end(n_reads=int(os.environ.get("CCSM_BENCH_READS", "8000")), read_len=15000)

    def call_mods_e2e_trained():
        # the same run on the committed trained checkpoint: what `call_mods --arithmetic auto` gives a user's model (split3)
        from ccsmeth_amd.utils import benchdata
        wt = dict(np.load(os.path.join(ROOT, "tests", "golden", "trained", "planted7_5000.npz")))
        r = benchdata.call_mods_end_to_end(n_reads=int(os.environ.get("CCSM_BENCH_READS", "8000")), read_len=15000, weights=wt)
        r["checkpoint"] = "tests/golden/trained/planted7_5000.npz (served in split3)"
        return r

    def aggregate():
        from ccsmeth_amd.utils import benchdata
        return benchdata.aggregate_50m(dev)

    def torch_cpu():
        # The same model assembled from stock PyTorch modules (nn.Embedding, nn.GRU, nn.Linear: what the reference's CPU path executes,
        # models.py:89-150 / utils/attention.py:48-70) on the host cores this container may use; neither oracle/ nor the reference is
        # involved.  Also a second, independent check of the HIP path's probabilities.
        import math
        from ccsmeth_amd.utils import synth
        threads = len(os.sched_getaffinity(0))
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if quota != "max":
                threads = min(threads, max(1, math.ceil(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
        torch.set_num_threads(threads)
        tw = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()}
        gru = torch.nn.GRU(11, 256, 3, batch_first=True, bidirectional=True)
        gru.load_state_dict({k[4:]: v for k, v in tw.items() if k.startswith("rnn.")})
        gru.eval()

        def strand(kmer, ipd, pw, npass, h0):
            x = torch.cat([tw["embed.weight"][kmer.long()], ipd[..., None], pw[..., None], npass[:, None, None].expand(-1, 21, 1)], 2)
            out, hn = gru(x, h0)
            q = torch.cat([hn[-2], hn[-1]], 1) @ tw["_att3.Wa.weight"].T
            e = torch.tanh(q[:, None, :] + out @ tw["_att3.Ua.weight"].T) @ tw["_att3.va.weight"].T
            return (torch.softmax(e, 1) * out).sum(1)

        def forward(s, h1, h2):
            t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in s.items()}
            with torch.no_grad():
                c = torch.cat([strand(t["kmer1"], t["ipd1"], t["pw1"], t["npass1"], torch.from_numpy(h1)),
                               strand(t["kmer2"], t["ipd2"], t["pw2"], t["npass2"], torch.from_numpy(h2))], 1)
                return torch.softmax(c @ tw["fc1.weight"].T + tw["fc1.bias"], 1).numpy()
        n = 512 * max(1, threads // 4)
        s = synth.synth_sites(n, 991)
        h1, h2 = synth.synth_h0(n, 992)
        forward(s, h1, h2)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 6.0:
            probs = forward(s, h1, h2)
            reps += 1
        dt = time.perf_counter() - t0
        ws = dm.workspace(n)
        _, gpu = ws.forward_host(s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h0=(h1, h2))
        ws.close()
        return {"value": reps * n / dt, "unit": "sites/s", "cores": threads, "sites_per_call": n, "calls": reps,
                "gpu_prob_max_abs_err_vs_torch_cpu": float(np.abs(gpu - probs).max()),
                "what": "stock PyTorch %s CPU modules (nn.GRU 3 x bidirectional + attention + fc), fp32, torch.set_num_threads(%d), explicit h0"
                        % (torch.__version__, threads)}

    leg("trained", trained)
    for prec, name in ((3, "split3"), (6, "split-mx-d"), (5, "hybrid"), (4, "split-mx")):     # the arithmetics the rule did not select for these weights
        if prec != dm.precision:
            leg(name, other_arithmetic(prec))
    leg("torch_cpu_path", torch_cpu)
    leg("pcie_inclusive", pcie)
    leg("call_mods_end_to_end", call_mods_e2e)
    leg("call_mods_end_to_end_trained", call_mods_e2e_trained)
    leg("aggregate_50M", aggregate)
    # BASELINE configs[2] at 1/10 of its size, measured once per round on the GPU box by tools/e2e_million_reads.py (8 minutes: not
    # re-run by this command; CCSM_BENCH_MILLION_READS=1 runs it here): the committed trained checkpoint, i.e. split3
    if os.environ.get("CCSM_BENCH_MILLION_READS"):
        def million():
            import subprocess
            logp = os.path.join(ROOT, "gpurun_out", "bench_million_reads.log")
            os.makedirs(os.path.dirname(logp), exist_ok=True)
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "e2e_million_reads.py"), "--log", logp], stdout=subprocess.DEVNULL)
            rep = [ln for ln in open(logp) if ln.startswith("# report:")][-1]
            return {"value": float(rep.split("->")[1].split("M sites/s")[0]) * 1e6, "unit": "sites/s", "log": logp, "what": rep.strip()}
        leg("call_mods_million_reads", million)
    else:
        out["call_mods_million_reads"] = {
            "value": None, "unit": "sites/s", "measured": "not by this command (CCSM_BENCH_MILLION_READS=1 runs it here: ~8 minutes)",
            "cited": {"sites_per_s": 1.681e6, "round": 5, "log": "profiles/r05_y_call_mods_million_reads_trained.log",
                      "what": "python -m ccsmeth_amd call_mods --io native --no_sort on 1 008 000 synthetic 15-kb HiFi reads (52.8 GiB of BGZF, 760.7 M CpG sites; "
                              "BASELINE configs[2] names 10 M reads: scaled down 9.9 x) with tests/golden/trained/planted7_5000.npz (split3): work phase 452.6 s"}}
    return out


def call_mods_multi_gpu(nranks, reads_per_rank=24000, timeout_s=900):
    """`python -m torch.distributed.run --nproc-per-node N -m ccsmeth_amd call_mods` on a synthetic HiFi BAM (8000 generated reads, their
    records 3 N times over: 18 M sites and 1.25 GiB per GPU), one process per GPU: BASELINE configs[3] scaled to what a bench run may take
    (~20 s of generation, ~10 s of run)."""
    import shutil
    import tempfile
    import torch
    from collections import OrderedDict
    from ccsmeth_amd.utils import benchdata, synth
    tmp = tempfile.mkdtemp(prefix="ccsm_bench_mg_")
    try:
        base, inp, ckpt, rep = (os.path.join(tmp, f) for f in ("base.bam", "in.bam", "m.ckpt", "report.json"))
        gen_s, _ = benchdata.write_synthetic_hifi_bam(base, 8000, 15000)
        times = max(1, reads_per_rank * nranks // 8000)
        size = benchdata.replicate_bam(base, inp, times)
        os.remove(base)
        torch.save(OrderedDict((k, torch.from_numpy(v)) for k, v in synth.synth_weights(5).items()), ckpt)
        cores = len(os.sched_getaffinity(0))
        try:
            q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                cores = min(cores, max(1, int(int(q) / int(p))))
        except (OSError, ValueError):
            pass
        threads = max(2, min(16, cores // nranks))
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        drop = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "ROLE_NAME",
                "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS")
        env = {k: v for k, v in os.environ.items() if k not in drop and not k.startswith(("TORCHELASTIC_", "TORCH_NCCL_", "CCSM_BENCH_"))}
        env.update(CCSM_CALLMODS_REPORT=rep, HSA_ENABLE_IPC_MODE_LEGACY="0",
                   PYTHONPATH=os.path.dirname(os.path.abspath(__file__)) + os.pathsep + env.get("PYTHONPATH", ""))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1",
               "--master-port", str(port), "-m", "ccsmeth_amd", "call_mods", "-i", inp, "-m", ckpt, "-o", os.path.join(tmp, "out"),
               "--batch_size", "12288", "--threads", str(threads)]
        t0 = time.time()
        p = subprocess.run(cmd, env=env, cwd=os.path.dirname(os.path.abspath(__file__)), capture_output=True, text=True, timeout=timeout_s)
        wall = time.time() - t0
        if p.returncode != 0 or not os.path.exists(rep):
            return {"error": "rc %d: %s" % (p.returncode, (p.stderr or "")[-600:])}
        d = json.load(open(rep))
        work = max(d.get("rank_seconds_work", [d.get("seconds_work", 0.0)]))
        return {"value": d["sites"] / d["seconds"], "unit": "sites/s", "ranks": nranks, "reads": d["reads"], "sites": d["sites"], "input_GiB": size / 2 ** 30,
                "seconds": d["seconds"], "seconds_work_slowest_rank": work, "work_phase_sites_per_s": d["sites"] / max(work, 1e-9),
                "seconds_stitch": d.get("seconds_stitch"), "seconds_index": d.get("seconds_index"), "rank_chunks": d.get("rank_chunks"),
                "threads_per_rank": threads, "host_cores": cores, "process_wall_s": wall,
                "what": "BASELINE configs[3] scaled down: python -m torch.distributed.run --nproc-per-node %d -m ccsmeth_amd call_mods on a synthetic "
                        "BAM (8000 generated 15-kb reads x %d), one process per GPU, chunk queue, one stitched modbam + index; `value` = sites / "
                        "in-run seconds incl. model set-up and the gather / stitch / index tail, `work_phase_sites_per_s` = sites / the slowest "
                        "rank's reading-to-writing phase" % (nranks, times)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(a)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; launch with --nproc-per-node == --gpus\n" % (a.gpus, world))
        sys.exit(2)
    if torch.cuda.device_count() == 1 and local_rank > 0:
        # a launcher that gives every rank a visible device of its own (HIP_VISIBLE_DEVICES per rank): the one visible device is index 0
        # (ranks that SHARE a device this way are what the census below refuses)
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        sys.stderr.write("bench.py: rank %d has no GPU (device_count %d)\n" % (local_rank, torch.cuda.device_count()))
        sys.exit(2)
    use_dist = world > 1 or os.environ.get("CCSM_BENCH_FORCE_DIST") == "1"     # the variable: exercise the RCCL path on a one-GPU box
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # RCCL's version banner goes to STDOUT (this image exports NCCL_DEBUG=VERSION): the line below must stay the only thing there
        os.environ["NCCL_DEBUG"] = os.environ.get("CCSM_BENCH_NCCL_DEBUG", "WARN")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        dist.barrier()              # pays the communicator's lazy set-up (hundreds of ms of idle GPU) HERE, not in the fence in front of the timed region
    n_gpus = world
    # N ranks are N GPUs only if they sit on N distinct physical devices: every rank reports what names its device, and the line is refused
    # (rc 2) when fewer distinct devices than ranks answer (CCSM_BENCH_ALLOW_SHARED_DEVICE=1: tests of the path on a one-GPU box)
    from ccsmeth_amd import sharding
    idents = [sharding.device_identity(local_rank)]
    if use_dist:
        gathered = [None] * world
        dist.all_gather_object(gathered, idents[0])
        idents = gathered
    census = sharding.device_census(idents, world)
    census["collective_library"] = sharding.collective_library() if use_dist else None
    if not census["ok"] and os.environ.get("CCSM_BENCH_ALLOW_SHARED_DEVICE") != "1":
        if rank == 0:
            sys.stderr.write("bench.py: %d rank(s) but %d answered from %d distinct device(s) %s; refusing to report a %d-GPU number\n"
                             % (world, census["ranks_seen"], census["distinct_devices"], census["devices"], world))
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(2)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from ccsmeth_amd.models import DeviceModel
    from ccsmeth_amd.utils import synth
    weights = dict(np.load(a.weights)) if a.weights else synth.synth_weights(20260928)
    dm = DeviceModel(weights, device=local_rank, precision=a.precision)

    # synthetic site pool resident in HBM: 8 distinct 2048-site batches per rank, cycled (SURVEY.md 8(d) generator)
    npool = 8
    sites = synth.synth_sites(BATCH * npool, 20260928 + 1000 * rank)
    pool = []
    for b in range(npool):
        sl = slice(b * BATCH, (b + 1) * BATCH)
        pool.append(tuple(torch.from_numpy(np.ascontiguousarray(sites[k][sl])).to(dev) for k in
                          ("kmer1", "ipd1", "pw1", "npass1", "kmer2", "ipd2", "pw2", "npass2")))
    grp = max(1, a.coalesce)
    runner = Runner(dm, pool, dev, grp, rank)

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    elapsed, full, rag, w_steps = timed(runner, a.steps, a.warmup, fence)
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel launch durations: mean over the FULL groups of the timed region (HIP events recorded on the stream each
    # kernel was launched on; the warm-up runs were dropped by re-arming the timers after the warm-up fence); a ragged group has
    # its own workspace and timers
    kt, nruns = runner.kernel_times(full_groups=bool(full))
    kt = np.array(kt)
    dom_ms = float(kt[1:3].mean())
    sites_per_launch = BATCH * (grp if full else rag)
    assert bool(torch.isfinite(runner.outs[0][0][1]).all())

    # the same loop over 240 steps (40 full groups, ~0.2 s), behind the contract's K steps: a 20-step region is 17 ms on a chip whose clock
    # follows its power over tens of milliseconds ("thin"); reported beside the headline, never instead of it
    sustained = None
    if n_gpus == 1 and a.steps < 240:
        s_steps = 40 * grp
        s_dt, _, _, _ = timed(runner, s_steps, 0, fence)
        sustained = {"value": s_steps * BATCH / s_dt, "unit": "sites/s", "steps": s_steps, "ms_per_step": s_dt / s_steps * 1e3,
                     "what": "the timed loop again over %d steps (full groups only), right behind the %d-step region" % (s_steps, a.steps)}
    if rank == 0:
        value = n_gpus * a.steps * BATCH / elapsed
        dtype, arith, passes = ARITH[dm.precision]
        achieved = 2.0 * MAC_GRU12 * sites_per_launch / (dom_ms * 1e-3)
        mx16 = dm.precision == 4 and bool(os.environ.get("CCSM_MX_SHAPE16"))    # opt-in: plain split-mx's layers 1-2 on the 16-wide instructions (ccsm_gru_mx16.hip)
        dom_kernel = ("gru_layer12_mx16_kernel" if mx16 else "gru_layer12_mx_kernel" if dm.precision >= 4 else
                      "gru_layer12_f3_kernel" if os.environ.get("CCSM_F3_SHAPE32") else "gru_layer12_f3s_kernel")
        traffic, traffic_src = traffic_of(dm.precision, dom_kernel)
        if mx16:
            arith = arith.replace("v_mfma_f32_32x32x16_f16 + (lo*hi, hi*lo) on v_mfma_scale_f32_32x32x64_f8f6f4",
                                  "v_mfma_f32_16x16x32_f16 + (lo*hi, hi*lo) of two pairs of k-blocks on v_mfma_scale_f32_16x16x128_f8f6f4 in layers 1-2 "
                                  "(layer 0 and the attention pool: v_mfma_f32_32x32x16_f16 + v_mfma_scale_f32_32x32x64_f8f6f4)")
        line = {
            "metric": "CpG sites/sec (call_mods, attbigru2s b21)", "value": value, "unit": "sites/s",
            "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": "attbigru2s_b21 forward on synthetic 21-mer CpG batches (BASELINE.json configs[1])",
                       "batch": BATCH, "sites_per_step": BATCH, "coalesce": grp, "streams": 2 if runner.overlap else 1, "full_groups": full, "ragged_group_batches": rag,
                       "warmup_steps_run": w_steps, "h0": "device Philox N(0,1)", "arithmetic": arith,
                       "arithmetic_selected": ARITH_NAME.get(dm.precision, dm.precision), "probe_max_abs_dprob": dm.probe_error,
                       "probe_q999_abs_dprob": dm.probe_q999, "probe_sites": dm.probe_sites,
                       "weights": ("state dict from %s" % a.weights) if a.weights else
                                  "synthetic random initialisation (seed 20260928), as BASELINE.json configs[1] defines the benchmark.  NOT what a trained "
                                  "checkpoint is served with: see trained_checkpoint below (ccsm_create serves trained weights in split3)",
                       "trained_checkpoint": "not measured (--extras none or N > 1)",
                       "parallelism": "reads sharded per GPU, no collective" if n_gpus > 1 else "single GPU",
                       "ranks_seen": census["ranks_seen"], "distinct_devices": census["distinct_devices"], "devices": census["devices"],
                       "collective_library": census["collective_library"]},
            "roofline": {"bound": "mfma", "kernel": dom_kernel + " (BiGRU layers 1-2)",
                         "achieved": achieved / 1e12, "peak": PEAK_F16_MFMA / 1e12, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F16_MFMA,
                         "traffic": None if traffic is None else traffic * sites_per_launch,
                         "traffic_source": None if traffic is None else traffic_src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, scaled per site)",
                         "launch_ms": dom_ms, "mfma_passes_per_flop": passes,
                         "issued_frac": achieved * passes / PEAK_F16_MFMA,
                         "power_note": "the kernel runs at the package power cap (sclk ~1.65-1.75 GHz of 2.4): profiles/r02_c_power_attribution.md; see peak_power_capped",
                         "note": "achieved = algorithmic flops of one launch (%d sites x 99.09 MFLOP) / its HIP-event duration; one launch = "
                                 "%d workgroups of 96 strand rows on 256 CUs" % (sites_per_launch, 2 * ((2 * int(sites_per_launch) + 95) // 96)),
                         "hbm_algorithmic_GBps": value / n_gpus * BYTES_PER_SITE / 1e9,
                         "hbm_frac": value / n_gpus * BYTES_PER_SITE / PEAK_HBM},
            "kernel_ms": {"gru0": float(kt[0]), "gru1": float(kt[1]), "gru2": float(kt[2]), "attn_fc": float(kt[3]),
                          "finalize": float(kt[4]), "launches_averaged": int(nruns)},
            "whole_path_TFLOPs": value / n_gpus * FLOP_PER_SITE / 1e12,
        }
        if sustained is not None:
            line["sustained"] = sustained
            line["value_sustained_240"] = sustained["value"]
        # the same device's MFMA ceiling under its power cap, measured live (3 s each; after the timed region): the GRU kernels'
        # instruction mix with random register-resident operands, and the same with the activations re-read from LDS
        if a.ceiling_seconds > 0:
            import ctypes as C
            from ccsmeth_amd import _lib as L
            capped = {}
            for mode, key in ((1, "mix"), (2, "mix_lds"), (0, "f16_32x32x16"), (3, "f16_16x16x32"), (4, "mix_16wide")):
                tf, gc = C.c_float(0), C.c_float(0)
                L.check(dm._lib.ccsm_measure_mfma_ceiling(local_rank, mode, float(a.ceiling_seconds) if mode in (1, 2) else min(2.0, float(a.ceiling_seconds)),
                                                          C.byref(tf), C.byref(gc)))
                capped[key] = (float(tf.value), float(gc.value))
            rl = line["roofline"]
            # the ceiling depends on the instruction's SHAPE.  split3's GRU layers issue v_mfma_f32_16x16x32_f16 (three passes per flop);
            # plain split-mx issues the 32-wide mix (its 16-wide form, round 6, is opt-in: CCSM_MX_SHAPE16=1)
            rl["peak_power_capped_by_shape"] = {"f16_32x32x16": capped["f16_32x32x16"][0], "f16_16x16x32": capped["f16_16x16x32"][0],
                                                "split_mx_mix_32wide": capped["mix"][0], "split_mx_mix_16wide": capped["mix_16wide"][0],
                                                "unit": "fp16-MFMA TFLOP/s, random register-resident operands"}
            if dm.precision == 3:
                f16c = capped["f16_32x32x16" if os.environ.get("CCSM_F3_SHAPE32") else "f16_16x16x32"][0]
                rl["frac_of_capped_split3"] = achieved * passes / 1e12 / f16c
                rl["frac_of_capped_split3_note"] = "achieved x 3 passes per flop / the fp16 ceiling of the instruction this kernel issues"
            mixkey = "mix_16wide" if mx16 else "mix"                # the ceiling of the instruction mix the dominant kernel issues
            rl["peak_power_capped"] = capped[mixkey][0]
            rl["frac_of_capped"] = achieved / 1e12 / capped[mixkey][0]
            rl["peak_power_capped_mix"] = "split_mx_mix_16wide" if mx16 else "split_mx_mix_32wide"
            # how far the north star's 5 M sites/s is from physics: the whole path's 244.23 MFLOP per site at the capped ceiling of the split-mx
            # mix (every cycle of every SIMD an MFMA of the kernel's mix, nothing else drawing power), and of three fp16 passes for split3
            rl["ceiling_sites_per_s"] = {"split_mx_at_capped_mix_16wide": capped["mix_16wide"][0] * 1e12 / FLOP_PER_SITE,
                                         "split_mx_at_capped_mix_32wide": capped["mix"][0] * 1e12 / FLOP_PER_SITE,
                                         "split3_at_capped_f16_16x16x32": capped["f16_16x16x32"][0] * 1e12 / 3.0 / FLOP_PER_SITE,
                                         "at_datasheet_f16_peak_one_pass": PEAK_F16_MFMA / FLOP_PER_SITE,
                                         "note": "capped ceiling (fp16-MFMA TFLOP/s; the correction products are overhead) / 244.23 MFLOP per site"}
            rl["peak_power_capped_lds_fed"] = capped["mix_lds"][0]
            rl["frac_of_capped_lds_fed"] = achieved / 1e12 / capped["mix_lds"][0]
            rl["power_capped_note"] = ("ccsm_measure_mfma_ceiling on this device, %.0f s per mode: one 512-thread workgroup per CU issuing the kernel's "
                                       "MFMA mix (per two v_mfma_f32_32x32x16_f16 one fp4 x fp6 v_mfma_scale_f32_32x32x64_f8f6f4) on RANDOM register-"
                                       "resident operands sustains %.0f TFLOP/s of fp16-MFMA flops at %.2f G issue cycles/s per SIMD (of 2.4), %.0f with "
                                       "the B operands re-read from LDS; full sweep with sclk / W: profiles/r03_a_power_ceiling.log"
                                       % (a.ceiling_seconds, capped["mix"][0], capped["mix"][1], capped["mix_lds"][0]))
        if n_gpus == 1 and a.cpu_seconds > 0:
            try:
                line["cpu_baseline"] = cpu_baseline(weights, a.cpu_seconds, dm)
            except ImportError as e:
                line["cpu_baseline"] = {"value": None, "unit": "sites/s", "cores": 0, "kind": "port", "sample": "unavailable: %s" % e}
        if n_gpus == 1 and a.extras == "all":
            line["extras"] = extras(weights, dm, dev, pool, grp)
            tr_ = line["extras"].get("trained", {})
            if "value" in tr_:      # next to the headline: the rate and arithmetic a TRAINED checkpoint gets (the headline is random-init weights)
                line["config"]["trained_checkpoint"] = {"arithmetic_selected": tr_["arithmetic_selected"], "sites_per_s": tr_["value"],
                                                        "max_abs_dprob_vs_oracle": tr_["max_abs_dprob_vs_oracle"], "checkpoint": tr_["checkpoint"]}
                line["value_trained_checkpoint"] = tr_["value"]
    runner.close()
    dm.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # N > 1: the line above is device-resident weak scaling; BASELINE configs[3] is the BAM -> modbam path on N GPUs.  With this
        # job's own ranks gone (they have nothing left to do), rank 0 runs that path once as its own torch.distributed.run job on a
        # synthetic BAM and reports it beside the headline - bounded by a timeout, never instead of the line
        legs = int(os.environ.get("CCSM_BENCH_MULTI_LEG_RANKS", n_gpus if n_gpus > 1 else 0))
        if legs > 1 and a.extras == "all":
            try:
                line.setdefault("extras", {})["call_mods_multi_gpu"] = call_mods_multi_gpu(legs)
            except Exception as e:      # noqa: BLE001
                line.setdefault("extras", {})["call_mods_multi_gpu"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:                        # whatever a native library still holds in C stdio comes out BEFORE the line: the line is the last one on stdout
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except (OSError, AttributeError):
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
