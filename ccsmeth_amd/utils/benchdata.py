"""Synthetic inputs and drivers of bench.py's secondary measurements (BASELINE.json configs[2] scaled down, configs[4])."""
import os
import tempfile
import time

import numpy as np


def write_synthetic_hifi_bam(path, n_reads, read_len, cpg=0.012, seed=3, planted=0.0):
    """Unaligned HiFi reads with kinetics tags (fi/fp/ri/rp uint8 per base, fn/rn passes): SURVEY.md 8(d)'s configs[2] shape,
    ~600 CpG sites per 15 kb read after the window filter.  planted > 0: every second read is "methylated" - the IPD / PW codes around
    each of its CpGs carry the shift pattern of synth.synth_labeled_sites (in units of the codes' standard deviation, x `planted`, a
    log-normal amplitude per site; forward strand around the C, reverse strand around the C opposite the G, x 0.7): kinetics with the
    structure a trained checkpoint responds to.  Returns (seconds, bytes)."""
    from .. import bamio
    rng = np.random.default_rng(seed)
    t0 = time.time()
    acgt = np.frombuffer(b"ACGT", np.uint8)
    pat_i = np.array([0.25, 0.55, 1.30, 0.80, 0.30]) * 28.0       # gamma(2, 20): sigma 28 codes
    pat_p = np.array([0.10, -0.20, 0.45, 0.30, 0.00]) * 28.0
    with bamio.BamWriter(path, "@HD\tVN:1.5\tSO:unknown\n", [], level=1) as w:
        for i in range(n_reads):
            seq = acgt[rng.choice(4, size=read_len, p=[0.3, 0.2, 0.2, 0.3])]
            pos = rng.integers(0, read_len - 1, int(read_len * cpg))
            seq[pos], seq[pos + 1] = ord("C"), ord("G")
            kin = np.clip(rng.gamma(2.0, 20.0, size=(4, read_len)), 0, 255).astype(np.uint8)
            if planted > 0 and i % 2 == 1:
                kf = kin.astype(np.float64)
                c = np.flatnonzero((seq[:-1] == ord("C")) & (seq[1:] == ord("G")))
                c = c[(c >= 12) & (c < read_len - 13)]
                amp = planted * np.exp(rng.normal(0.0, 0.45, len(c)))
                for d in range(5):
                    kf[0, c - 2 + d] += amp * pat_i[d]                          # fi / fp: the forward strand, around the C
                    kf[1, c - 2 + d] += amp * pat_p[d]
                    kf[2, read_len - 2 - c - 2 + d] += 0.7 * amp * pat_i[d]     # ri / rp (reverse-strand order): around the C opposite the G
                    kf[3, read_len - 2 - c - 2 + d] += 0.7 * amp * pat_p[d]
                kin = np.clip(np.rint(kf), 0, 255).astype(np.uint8)
            tags = [("fi", "BC", kin[0]), ("fp", "BC", kin[1]), ("ri", "BC", kin[2]), ("rp", "BC", kin[3]), ("fn", "C", 12), ("rn", "C", 13),
                    ("np", "C", 25)]
            w.write(bamio.BamRecord("m/%d/ccs" % i, flag=4, ref_id=-1, seq=seq.tobytes().decode(), qual=np.full(read_len, 40, np.uint8), tags=tags))
    return time.time() - t0, os.path.getsize(path)


def replicate_bam(src, dst, times, threads=8):
    """A `times` x larger copy of an unaligned BAM for throughput runs: the records are rewritten once behind a block-aligned
    header (native reader -> native writer, nothing changed), then that run of BGZF blocks is appended `times` times (read names
    repeat).  Returns the size of dst."""
    from .. import bamnative as bn
    tmp = dst + ".once"
    with bn.NativeBamReader(src, threads=threads) as rd, bn.NativeBamWriter(tmp, rd.header_text, rd.raw_refs, rd.n_ref, threads=threads, level=1) as w:
        header_end = w.flush()
        while True:
            b = rd.next_batch(512)
            if b is None:
                break
            w.write_batch(b, rm_pulse=False)
            b.close()
        end = w.flush()
    with open(tmp, "rb") as fh, open(dst, "wb") as out:
        out.write(fh.read(header_end))
        body = fh.read(end - header_end)
        for _ in range(times):
            out.write(body)
        out.write(bn.BGZF_EOF)
    os.remove(tmp)
    return os.path.getsize(dst)


def call_mods_end_to_end(n_reads=4000, read_len=15000, weights=None):
    """`call_mods --io native` on a synthetic BAM (BGZF inflate, parse, feature extraction + model on the GPU, MM/ML, BGZF deflate);
    second of two runs (the first pays page-ins and library loads).  weights: a state dict (default: the synthetic initialisation,
    which `--arithmetic auto` serves in split-mx; a trained checkpoint gets split3)."""
    import torch
    from collections import OrderedDict
    from ..call_mods import build_parser, call_mods
    from . import synth
    tmp = tempfile.mkdtemp(prefix="ccsm_bench_")
    inp = os.path.join(tmp, "in.bam")
    gen_s, nbytes = write_synthetic_hifi_bam(inp, n_reads, read_len)
    ckpt = os.path.join(tmp, "m.ckpt")
    torch.save(OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in (weights or synth.synth_weights(5)).items()), ckpt)
    res, dt = None, None
    for _ in range(2):
        args = build_parser().parse_args(["-i", inp, "-m", ckpt, "-o", os.path.join(tmp, "out"), "--batch_size", "12288", "--no_sort"])
        t0 = time.time()
        res = call_mods(args, log=open(os.devnull, "w"))
        dt = time.time() - t0
    sites = int(res.get("sites", 0))
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)
    return {"value": sites / dt, "unit": "sites/s", "reads": int(res["reads"]), "sites": sites, "seconds": dt, "input_MB": nbytes / 1e6,
            "what": "python -m ccsmeth_amd call_mods --io native --no_sort, wall time incl. model set-up; BASELINE configs[2] scaled from "
                    "10 M reads to %d synthetic %d-base reads (input generated in %.1f s)" % (n_reads, read_len, gen_s)}


def aggregate_50m(dev, regions=500, region_sites=100000):
    """BASELINE configs[4]: aggregate attbigru_b11 over 50 M pile-up sites, device-resident tables, per-region seeded h0 stream."""
    import ctypes as C  # noqa: F401
    import torch
    from .. import _lib
    from ..call_mods_freq_bam import AggrModel
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    w = dict(np.load(os.path.join(root, "tests", "golden", "aggr_ckpt_weights.npz")))
    model = AggrModel(w, device=dev.index or 0, stream_sites=region_sites)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    m = region_sites
    pos = torch.cumsum(torch.randint(2, 401, (m,), device=dev, generator=g), 0).to(torch.int64)
    cov = torch.randint(4, 61, (m, 1), device=dev, generator=g)
    # histogram of cov_i Beta(0.3, 0.3)-like calls per site: a U-shaped multinomial over 20 bins, L2-normalised, 6 decimals
    pbin = torch.tensor(np.diff(np.clip(np.linspace(0, 1, 21), 0, 1) ** 0.3) + np.diff(1 - (1 - np.linspace(0, 1, 21)) ** 0.3), device=dev, dtype=torch.float32)
    pbin = pbin / pbin.sum()
    draws = torch.multinomial(pbin.expand(m, -1), 60, replacement=True, generator=g)
    keep = (torch.arange(60, device=dev).expand(m, -1) < cov)
    hist = torch.zeros((m, 20), device=dev).scatter_add_(1, draws, keep.float())
    hist = torch.round(hist / hist.norm(dim=1, keepdim=True) * 1e6) / 1e6
    hist = hist.contiguous()
    out = torch.empty(m, dtype=torch.float32, device=dev)
    lib = model._lib
    st = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(3):
        _lib.check(lib.ccsm_aggr_forward_device(model.handle, m, pos.data_ptr(), hist.data_ptr(), 0, out.data_ptr(), st))
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(regions):
        _lib.check(lib.ccsm_aggr_forward_device(model.handle, m, pos.data_ptr(), hist.data_ptr(), 0, out.data_ptr(), st))
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    sites = regions * m
    ok = bool(torch.isfinite(out).all())
    return {"value": sites / dt, "unit": "sites/s", "sites": sites, "seconds": dt, "TFLOPs": sites / dt * 275.3e3 / 1e12,
            "algorithmic_GBps": sites / dt * 88 / 1e9, "finite": ok,
            "what": "aggregate attbigru_b11 (real checkpoint) over %d regions x %d sites, device-resident histograms and positions" % (regions, m)}
