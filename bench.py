#!/usr/bin/env python
"""bench.py — CpG sites/s of the MI355X-native attbigru2s call_mods hot path (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one forward of the hot path over one batch of 2048 synthetic CpG sites (both strands, 21-mers), features
already resident in HBM, initial states drawn on the device (Philox) as SURVEY.md 8(d) prescribes for the timed
run.  Steps are issued round-robin on `--streams` HIP streams (one workspace each) so that several batches are in
flight, exactly K steps are timed between barrier + synchronize on both sides, MAX over ranks is taken and rank 0
prints ONE JSON line.  Reads are sharded across GPUs with no collective on the data path (weak scaling).
"""
import argparse
import json
import os
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
# HBM/fabric bytes of ONE launch of the dominant kernel over 3 x 2048 sites, from rocprofv3 PMC passes on the same launch
# shape (profiles/r01_c_pmc_coalesced.md: 2 x FETCH_SIZE + WRITE_SIZE, KiB, gfx950 read correction per MI355X_MICROARCH.md).
# PMC counters cannot be read from inside this process; the figure is per-launch like `achieved` and scales with sites.
TRAFFIC_BYTES_PER_SITE_GRU12 = {3: (2 * 1241546 + 516096) * 1024 / 6144.0,      # profiles/r01_c_pmc_coalesced.md
                                4: (2 * 1252400 + 518150) * 1024 / 6144.0}      # profiles/r01_l_pmc_coalesced.md; other modes: null


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the coalesced groups alternate over (2 overlaps launch tails: +3 % value, but the overlapped\n"
                         "launches then report inflated per-kernel durations; 1 keeps roofline.achieved = a solo launch)")
    ap.add_argument("--coalesce", type=int, default=6,
                    help="batches run per launch of the heavy kernels (micro-batching).  6 x 2048 sites = 24576 strand rows = 512 GRU\n"
                         "workgroups (2 full rounds of the 256 CUs) and 768 attention workgroups (3 full rounds); 3 leaves the\n"
                         "attention kernel a half-empty second round (+20 %% on its per-site time)")
    ap.add_argument("--precision", type=int, default=4, choices=(1, 2, 3, 4),
                    help="4 = split-f8 (default: fp16 main product + fp8 correction products, max |dprob| ~4e-6); 3 = split-fp16\n"
                         "x3 (fp32-class, ~2e-7); 2/1 = fewer passes, outside the parity margin, reported as such")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline leg (0 = skip)")
    return ap.parse_args()


def cpu_baseline(weights, target_s, device_model):
    """The one leg that touches oracle/: oracle/attbigru2s_oracle.c on the host cores over a bounded sample of the same
    synthetic workload, plus (same leg, same probe inputs, h0 pinned) the max |delta prob| of the HIP path against it —
    the "prob delta vs ref" half of BASELINE.json's metric."""
    from ccsmeth_amd.utils import synth
    from oracle import c_oracle
    threads = c_oracle.max_threads()
    probe_n = 8 * threads
    s = synth.synth_sites(probe_n, 777)
    h1, h2 = synth.synth_h0(probe_n, 778)
    args = (s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h1, h2)
    c_oracle.forward(weights, *args)            # warm-up (thread pool, page faults)
    t0 = time.perf_counter()
    _, ref_probs = c_oracle.forward(weights, *args)
    rate = probe_n / (time.perf_counter() - t0)
    ws = device_model.workspace(probe_n)
    _, gpu_probs = ws.forward_host(s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"],
                                   h0=(h1, h2))
    ws.close()
    prob_err = float(np.abs(gpu_probs - ref_probs).max())
    n = int(min(max(rate * target_s, probe_n), 65536))
    n = (n // (8 * threads)) * 8 * threads or probe_n
    s = synth.synth_sites(n, 779)
    h1, h2 = synth.synth_h0(n, 780)
    t0 = time.perf_counter()
    c_oracle.forward(weights, s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h1, h2)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "sites/s", "cores": threads, "kind": "port",
            "sample": "%d synthetic sites (same generator as the GPU run), explicit h0, oracle/attbigru2s_oracle.c fp32 "
                      "AVX2+OpenMP, %.1f s" % (n, dt),
            "gpu_prob_max_abs_err": prob_err, "gpu_prob_err_sites": probe_n}


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert a.gpus == world or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"
    n_gpus = world
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from ccsmeth_amd.models import DeviceModel
    from ccsmeth_amd.utils import synth
    weights = synth.synth_weights(20260928)
    dm = DeviceModel(weights, device=local_rank, precision=a.precision)

    # synthetic site pool resident in HBM: 8 distinct 2048-site batches per rank, cycled (SURVEY.md 8(d) generator)
    npool = 8
    sites = synth.synth_sites(BATCH * npool, 20260928 + 1000 * rank)
    pool = []
    for b in range(npool):
        sl = slice(b * BATCH, (b + 1) * BATCH)
        pool.append(tuple(torch.from_numpy(np.ascontiguousarray(sites[k][sl])).to(dev) for k in
                          ("kmer1", "ipd1", "pw1", "npass1", "kmer2", "ipd2", "pw2", "npass2")))
    # Micro-batch coalescing (ccsm_group_*): every step binds its own 2048-site batch (own outputs, own initial states)
    # to a workspace; after `--coalesce` steps the heavy kernels run ONCE over those batches, so that one launch fills the
    # chip (3 x 2048 sites = 256 workgroups of 96 strand rows = one per CU; the default 6 = two full rounds).  Workspaces
    # alternate over `--streams` streams.
    nst = max(1, a.streams)
    grp = max(1, a.coalesce)
    wss = [dm.workspace(BATCH * grp) for _ in range(nst)]
    streams = [torch.cuda.Stream(dev) for _ in range(nst)]
    outs = [[(torch.empty((BATCH, 2), device=dev), torch.empty((BATCH, 2), device=dev)) for _ in range(grp)] for _ in range(nst)]
    for w in wss:
        w.set_timing(True)       # HIP events around each kernel, on the stream the kernel runs on

    def step(i, last=False):
        k = (i // grp) % nst
        wss[k].group_add_torch(*pool[i % npool], stream=streams[k].cuda_stream, out=outs[k][i % grp], seed=1234,
                               offset=(rank * 10**9 + i * BATCH))
        if (i + 1) % grp == 0 or last:
            wss[k].group_run(stream=streams[k].cuda_stream)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(a.warmup):
        step(i, last=(i == a.warmup - 1))
    fence()
    for w in wss:
        w.set_timing(True)       # re-arm: average only the timed region's launches
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i, last=(i == a.steps - 1))
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel launch durations: mean over every run of the timed region (HIP events recorded on the stream each
    # kernel was launched on; the warm-up runs were dropped by re-arming the timers after the warm-up fence)
    tms = [w.timing_mean() for w in wss]
    nruns = sum(n for _, n in tms)
    kt = np.sum([np.array(t) * n for t, n in tms], axis=0) / max(nruns, 1)     # ms: gru0, gru1, gru2, attn, finalize
    dom_ms = float(kt[1:3].mean())
    launches = -(-a.steps // grp)                           # group runs in the timed region (the timers keep the last 128 of them)
    sites_per_launch = a.steps * BATCH / launches           # = BATCH * coalesce when steps is a multiple of it
    assert bool(torch.isfinite(outs[0][0][1]).all())

    if rank == 0:
        value = n_gpus * a.steps * BATCH / elapsed
        passes = {4: 2, 3: 3, 2: 2, 1: 1}[a.precision]     # MFMA issue cycles per algorithmic flop, in fp16-rate units
        achieved = 2.0 * MAC_GRU12 * sites_per_launch / (dom_ms * 1e-3)
        line = {
            "metric": "CpG sites/sec (call_mods, attbigru2s b21)", "value": value, "unit": "sites/s",
            "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {4: "f16+f8 split operands, f32 accumulate", 3: "f16x3 split operands, f32 accumulate",
                      2: "f16 weights x split-f16 activations, f32 accumulate", 1: "f16 operands, f32 accumulate"}[a.precision],
            "data": "synthetic",
            "config": {"workload": "attbigru2s_b21 forward on synthetic 21-mer CpG batches (BASELINE.json configs[1])",
                       "batch": BATCH, "sites_per_step": BATCH, "streams": nst, "coalesce": grp, "h0": "device Philox N(0,1)",
                       "arithmetic": {4: "hi*hi on v_mfma_f32_32x32x16_f16 + (lo*hi, hi*lo) with fp8 e4m3 operands on "
                                         "v_mfma_scale_f32_32x32x64_f8f6f4, one fp32 accumulator (GRU layers and attention pool)",
                                      3: "split-fp16 x3 MFMA (hi*hi+hi*lo+lo*hi), fp32 accumulate",
                                      2: "fp16 weights x split-fp16 activations (2 MFMA passes), fp32 accumulate",
                                      1: "fp16 operands (1 MFMA pass), fp32 accumulate"}[a.precision],
                       "parallelism": "reads sharded per GPU, no collective" if n_gpus > 1 else "single GPU"},
            "roofline": {"bound": "mfma", "kernel": ("gru_layer_f8_kernel<KX=32>" if a.precision == 4 else "gru_layer_v2_kernel<KX=32>") + " (BiGRU layers 1-2)",
                         "achieved": achieved / 1e12, "peak": PEAK_F16_MFMA / 1e12, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F16_MFMA,
                         "traffic": TRAFFIC_BYTES_PER_SITE_GRU12.get(a.precision, 0) * sites_per_launch or None,
                         "traffic_source": "profiles/r01_%s_pmc_coalesced.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on a 6144-site launch, "
                                           "scaled per site)" % ("l" if a.precision == 4 else "c"),
                         "launch_ms": dom_ms, "mfma_passes_per_flop": passes,
                         "issued_frac": achieved * passes / PEAK_F16_MFMA,
                         "note": "achieved = algorithmic flops of one launch (%d sites x 99.09 MFLOP) / its HIP-event "
                                 "duration; one launch = %d workgroups of 96 strand rows on 256 CUs; launches of %d streams "
                                 "may overlap" % (sites_per_launch, 2 * ((2 * int(sites_per_launch) + 95) // 96), nst),
                         "hbm_algorithmic_GBps": value / n_gpus * BYTES_PER_SITE / 1e9,
                         "hbm_frac": value / n_gpus * BYTES_PER_SITE / PEAK_HBM},
            "kernel_ms": {"gru0": float(kt[0]), "gru1": float(kt[1]), "gru2": float(kt[2]), "attn_fc": float(kt[3]),
                          "finalize": float(kt[4]), "launches_averaged": int(nruns)},
            "whole_path_TFLOPs": value / n_gpus * FLOP_PER_SITE / 1e12,
        }
        if n_gpus == 1 and a.cpu_seconds > 0:
            try:
                line["cpu_baseline"] = cpu_baseline(weights, a.cpu_seconds, dm)
            except ImportError as e:
                line["cpu_baseline"] = {"value": None, "unit": "sites/s", "cores": 0, "kind": "port", "sample": "unavailable: %s" % e}
        print(json.dumps(line), flush=True)
    for w in wss:
        w.close()
    dm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
