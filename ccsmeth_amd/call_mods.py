"""`call_mods` driver: HiFi BAM with kinetics -> modbam, one GPU, no pysam.

Mirrors reference call_modifications.py:474-613 (argument checks, `<output>.modbam.bam`, @PG line, "bad read => written
untagged", end-of-run counters) with one process instead of the reference's reader/extract/call/writer process pool:
default (`--io native`): libccsm_bam (threaded BGZF, include/ccsm_bam.h) -> chunks of `--holes_batch` reads ->
ccsm_forward_reads_host (feature extraction + model on the GPU) -> libccsm_bam (tag refill, MM/ML, threaded BGZF);
`--io python`: bamio.BamReader -> pipeline.CallModsPipeline -> bamio.BamWriter, record by record.
Output order = input order; unless `--no_sort`, the file is then coordinate-sorted if it is not already in order and indexed
(.bai), the reference's samtools post-processing."""
import argparse
import os
import sys
import time

import numpy as np

from . import __version__
from ._bam2modbam import _convert_locs_to_mmtag, _convert_probs_to_mltag, _refill_tags
from .bamio import BamReader, BamWriter, add_pg_line
from .pipeline import CallModsPipeline, Read

REF_VERSION = "0.5.0"     # VN the reference writes (ccsmeth/_version.py)


def build_parser():
    """Flags, short options and defaults of the reference's `ccsmeth call_mods` (ccsmeth.py:196-326).  Flags whose non-default
    values select paths outside this build (align mode, transformer models, TSV input, ...) are accepted and
    rejected with a ValueError in call_mods(), like the reference rejects an unknown --model_type."""
    p = argparse.ArgumentParser("ccsmeth_amd call_mods", description="call 5mCpG from a HiFi BAM with kinetics (MI355X)")
    p.add_argument("--input", "-i", required=True, help="input BAM (fi/ri/fp/rp/fn/rn tags)")
    p.add_argument("--holes_batch", type=int, default=50,
                   help="reads per unit of work (reference default 50).  With --io native the value 50 is run as 256 (logged; a hole-batch\n"
                        "is cut into GPU launches of >= 12288 sites, its last launch is ragged and the GPU drains between hole-batches:\n"
                        "16000 reads run at 1.89 M sites/s with 64, 2.00-2.01 M with 128-512, 1.85 M with 2048); --holes_batch_exact keeps\n"
                        "50; any other value is taken as given.  The calls do not depend on it")
    p.add_argument("--output", "-o", required=True, help="output prefix; writes <output>.modbam.bam")
    p.add_argument("--gzip", action="store_true", default=False, help="(reference: TSV output only) ignored")
    p.add_argument("--keep_pulse", action="store_true", default=False)
    p.add_argument("--no_sort", action="store_true", default=False,
                   help="skip the post-processing (coordinate sort when the records are not in order, then the .bai index)")
    p.add_argument("--model_file", "-m", required=True, help=".ckpt (torch state_dict) of attbigru2s")
    p.add_argument("--model_type", default="attbigru2s")
    p.add_argument("--seq_len", type=int, default=21)
    p.add_argument("--is_npass", default="yes")
    p.add_argument("--is_stds", default="no")
    p.add_argument("--is_sn", default="no")
    p.add_argument("--is_map", default="no")
    p.add_argument("--class_num", type=int, default=2)
    p.add_argument("--dropout_rate", type=float, default=0)
    p.add_argument("--batch_size", "-b", type=int, default=512)
    p.add_argument("--layer_rnn", type=int, default=3)
    p.add_argument("--hid_rnn", type=int, default=256)
    p.add_argument("--layer_trans", type=int, default=6)
    p.add_argument("--nhead", type=int, default=4)
    p.add_argument("--d_model", type=int, default=256)
    p.add_argument("--dim_ff", type=int, default=512)
    p.add_argument("--mode", default="denovo")
    p.add_argument("--holeids_e", default=None)
    p.add_argument("--holeids_ne", default=None)
    p.add_argument("--motifs", default="CG")
    p.add_argument("--mod_loc", type=int, default=0)
    p.add_argument("--methy_label", type=int, default=1, choices=[1, 0])
    p.add_argument("--norm", default="zscore")
    p.add_argument("--no_decode", action="store_true", default=False)
    p.add_argument("--no_data_probe", action="store_true", default=False,
                   help="--arithmetic auto picks split-mx only for a checkpoint whose 65536-site SYNTHETIC probe is clean; call_mods then\n"
                        "repeats the comparison against split3 on the first <= 65536 sites of THIS input and falls back to split3 if\n"
                        "split-mx is not clean on them as well (costs one extra pass over those sites).  This flag skips that second probe.")
    p.add_argument("--shadow_every", type=int, default=64,
                   help="with split-mx served behind both probes: one chunk in this many is also computed in split3 and compared (max |dprob|\n"
                        "<= 1.25e-5: the rule's first condition), so that the rule covers the whole input, not its first 65536 sites (~1.5 %% of\n"
                        "the run's model time at 64).  A violation starts the run again in split3 (single process) or stops every rank with the\n"
                        "message.  0 switches the shadow off.")
    p.add_argument("--ref", default=None)
    p.add_argument("--mapq", type=int, default=1)
    p.add_argument("--identity", type=float, default=0.0)
    p.add_argument("--no_supplementary", action="store_true", default=False)
    p.add_argument("--skip_unmapped", default="yes")
    p.add_argument("--threads", "-p", type=int, default=10, help="BGZF inflate / deflate threads of --io native")
    p.add_argument("--dispatch", default="dynamic", choices=("dynamic", "static"),
                   help="multi-GPU runs (torch.distributed.run): dynamic = every rank claims the next unclaimed chunk of the input (the "
                        "reference's shared queue); static = chunk i goes to rank i mod world")
    p.add_argument("--chunk_mb", type=float, default=32.0,
                   help="multi-GPU runs: compressed MiB of input per unit handed to a rank (a rank finds its chunk's first record itself; "
                        "nothing is inflated twice)")
    p.add_argument("--holes_batch_exact", action="store_true", default=False,
                   help="--io native: run --holes_batch 50 as 50 reads (by default the reference's 50 is run as 256: see --holes_batch)")
    p.add_argument("--threads_call", type=int, default=3, help="(reference: call workers) ignored: one process per GPU")
    p.add_argument("--tseed", type=int, default=1234)
    p.add_argument("--use_compile", default="no")
    # this build's own switches
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--io", default="native", choices=["native", "python"],
                   help="BAM reader/writer: libccsm_bam (threaded BGZF, whole read chunks straight to the GPU; implies --extract\n"
                        "device) or the pure-Python record-by-record implementation")
    p.add_argument("--arithmetic", default="auto", choices=["auto", "split3", "hybrid", "split-mx-d", "split-mx"],
                   help="MFMA arithmetic of the model (the reference computes in fp32).  auto (default): split3 (three fp16 passes on\n"
                        "hi + lo operands: fp32-class, max abs error < 1e-6) unless a 65536-site probe of THIS checkpoint against split3\n"
                        "finds split-mx (fp16 main product + one block-scaled fp6/fp4 correction product, 1.5x faster) within 1.25e-5 at\n"
                        "every site with a light tail - true of freshly initialised weights, not of trained checkpoints, whose split-mx /\n"
                        "split-mx-d / hybrid errors are heavy-tailed (single sites beyond 1e-4 among 10^6).  The other values force one\n"
                        "arithmetic: split-mx-d (fp6 recurrent weights, block-scaled state) and hybrid (three passes in the recurrent part)\n"
                        "trade that tail for 1.25-1.4x split3's speed, at the caller's risk")
    p.add_argument("--extract", default="device", choices=["device", "host"],
                   help="where the 21-mer features are built: on the GPU from the raw read arrays (default) or NumPy on the host")
    return p


def _check_scope(args):
    """Reject, loudly, the reference options this build does not implement (SURVEY.md 8 scope)."""
    yes = lambda v: str(v).lower() in ("yes", "true", "t", "1")  # noqa: E731  (ccsmeth str2bool)
    if args.model_type != "attbigru2s":
        raise ValueError("--model_type not right!")                    # call_modifications.py:340
    if not yes(args.is_npass) or yes(args.is_stds) or yes(args.is_sn) or yes(args.is_map):
        raise ValueError("this command builds the default per-site features (--is_npass yes --is_stds no --is_sn no --is_map no); the model "
                         "itself takes the other variants through ModelAttRNN.forward / ccsm_forward_host (include/ccsm.h)")
    if (args.layer_rnn, args.hid_rnn, args.class_num, args.seq_len) != (3, 256, 2, 21):
        if args.seq_len % 2 == 0:
            raise ValueError("--seq_len must be odd")                  # :500-501
        raise ValueError("this build implements --seq_len 21 --layer_rnn 3 --hid_rnn 256 --class_num 2")
    if args.mode not in ("denovo", "align"):
        raise ValueError("--mode must be denovo or align")
    if args.mode == "align" and args.ref is not None and not os.path.exists(args.ref):
        raise ValueError("--ref does not exist")       # only --is_map yes reads the reference sequence (outside this build)
    if args.motifs.upper() != "CG" or args.mod_loc != 0:
        raise ValueError("this build implements --motifs CG --mod_loc 0")
    if args.norm not in ("zscore", "min-mean", "min-max", "mad", "none"):
        raise ValueError("--norm must be one of zscore, min-mean, min-max, mad, none")         # the reference's choices (ccsmeth.py)
    if yes(args.use_compile):
        raise ValueError("--use_compile applies to the reference's torch model only")


def _set_coordinate_order(header_text):
    """@HD ... SO:coordinate, as `samtools sort` leaves the header."""
    lines = header_text.split("\n")
    if lines and lines[0].startswith("@HD"):
        f = [x for x in lines[0].split("\t") if not x.startswith("SO:")]
        lines[0] = "\t".join(f[:2] + ["SO:coordinate"] + f[2:]) if len(f) >= 2 else lines[0] + "\tSO:coordinate"
        return "\n".join(lines)
    return "@HD\tVN:1.6\tSO:coordinate\n" + header_text


def _post_sort_index(out_path, args, log):
    """call_modifications.py:592-607: unless --no_sort, `samtools sort` then `samtools index` of the modbam; failures are
    warnings there too.  Records already in coordinate order (every unaligned HiFi BAM: the order is kept) are only indexed."""
    if args.no_sort:
        return
    from . import bamnative
    t = time.time()
    try:
        did = bamnative.sort_and_index(out_path, threads=max(1, args.threads), max_bytes=_sort_limit())
        print("[post_process] bam_sort_index costs %.2f seconds (%s)" % (time.time() - t, "sorted + indexed" if did else "already in order: indexed"),
              file=log)
    except Exception as e:  # noqa: BLE001
        # the modbam itself is complete; tell the user exactly what is missing (the sort spills to disk above its memory limit, so this
        # is a full disk, an unreadable file or the like)
        print("[post_process] WARNING: sorting / indexing %s FAILED (%s): the file is unsorted and has no index; run "
              "`samtools sort` + `samtools index` on it before call_freqb" % (out_path, e), file=log)
        return False
    return True


def _sort_limit():
    try:
        return int(os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") * 0.5)
    except (ValueError, OSError):
        return 0


def _align_skip_and_window(flag, mapq, identity, qstart, qend, length, args):
    """--mode align (extract_features.py:272-304, 383-391): reads that are unmapped / secondary / duplicate, supplementary with
    --no_supplementary, below --mapq or below --identity give no features; with --skip_unmapped yes a site is kept only inside
    the aligned part of the read, [seq_start, seq_end) in forward-sequence coordinates (flipped for reverse-strand records)."""
    flag, mapq = np.asarray(flag), np.asarray(mapq)
    skip = (flag & (0x4 | 0x100 | 0x400)) != 0
    if args.no_supplementary:
        skip |= (flag & 0x800) != 0
    skip |= mapq < args.mapq
    skip |= np.asarray(identity) < args.identity
    if str(args.skip_unmapped).lower() not in ("yes", "true", "t", "1"):
        return skip, None
    rev = (flag & 16) != 0
    lo = np.where(rev, length - qend, qstart).astype(np.int64)
    hi = np.where(rev, length - qstart, qend).astype(np.int64)
    return skip, (lo, hi)


def _cigar_align_info(cigar, l_seq):
    """(query_alignment_start, query_alignment_end, identity) of a record's CIGAR [(op, len)], as ccsm_bam_align_info."""
    qs, qe = 0, l_seq
    for op, ln in cigar:
        if op == 5:
            continue
        if op != 4:
            break
        qs += ln
    for op, ln in reversed(cigar):
        if op == 5:
            continue
        if op != 4:
            break
        qe -= ln
    cnt = [0] * 16
    for op, ln in cigar:
        cnt[op] += ln
    nalign = sum(cnt[i] for i in (0, 1, 2, 3, 6, 7, 8, 9))
    return qs, qe, ((cnt[0] + cnt[7]) / float(nalign) if nalign > 0 else 0.0)


def _filter_sites_by_window(first, locs, prob1, tagged, window):
    """Drop the sites outside each read's [lo, hi): same arrays, compacted."""
    lo, hi = window
    n = len(first) - 1
    cnt = np.diff(first)
    rid = np.repeat(np.arange(n), cnt)
    keep = (locs >= lo[rid]) & (locs < hi[rid])
    newcnt = np.bincount(rid[keep], minlength=n).astype(np.int32)
    nf = np.zeros(n + 1, np.int32)
    np.cumsum(newcnt, out=nf[1:])
    return nf, np.ascontiguousarray(locs[keep]), np.ascontiguousarray(prob1[keep]), (np.asarray(tagged, bool) & (newcnt > 0)).astype(np.uint8)


def _get_holes(holeidfile):
    """extract_features.py:76-84: one read name per line."""
    holes = set()
    with open(holeidfile, "r") as rf:
        for line in rf:
            if line.strip():
                holes.add(line.strip())
    return holes


def _skip_by_name(name, holeids_e, holeids_ne):
    """extract_features.py:268-271: reads outside --holeids_e or inside --holeids_ne yield no features (written untagged)."""
    return (holeids_e is not None and name not in holeids_e) or (holeids_ne is not None and name in holeids_ne)


def _batch_names(batch):
    """Read names of a bamnative.Batch (BAM record: block_size i32, 32 fixed bytes with l_read_name at +8, then the name)."""
    names = []
    rec = batch.records
    for r in range(batch.n_reads):
        o = int(batch.rec_offset[r]) + 4
        l_name = int(rec[o + 8])
        names.append(bytes(rec[o + 32:o + 32 + l_name - 1]).decode("ascii"))
    return names


def _load_state_dict(path):
    import torch
    sd = torch.load(path, map_location="cpu")
    return sd


def _read_of(rec):
    def tag(name, default):
        try:
            return rec.get_tag(name)
        except KeyError:
            return default
    fwd = rec.get_forward_sequence()
    return Read(rec.query_name, fwd, np.asarray(tag("fi", [])), np.asarray(tag("ri", [])), np.asarray(tag("fp", [])),
                np.asarray(tag("rp", [])), tag("fn", 0), tag("rn", 0), rec.is_reverse)


def _print_data_probe(dm, log):
    print("[main]arithmetic on this input: split-mx against split3 over its first %d sites: max %.1e, 99.9 %% %.1e -> %s" % (
        dm.data_probe_sites, dm.data_probe_error, dm.data_probe_q999,
        "split-mx is served" if dm.precision == 4 else "NOT clean on this input: split3 (three fp16 passes, fp32-class) is served"), file=log)


def call_mods(args, log=sys.stderr, pipe=None):
    """`pipe`: an object with CallModsPipeline's run_native_batch / close (the CPU tests of the multi-rank hand-out pass a
    stand-in; None = the GPU pipeline on the checkpoint of --model_file).

    --arithmetic auto serves split-mx only behind two probes (the checkpoint on synthetic sites, then the input's first <= 65536 sites) AND
    a running shadow: one chunk in --shadow_every is also computed in split3 and compared under the rule's first condition.  A violation
    anywhere in the file aborts the run, removes what it wrote and starts again in split3 (single process; several ranks: every rank stops
    with the message): the output is then byte-identical to --arithmetic split3."""
    from .pipeline import ArithmeticViolation
    try:
        return _call_mods_once(args, log, pipe)
    except ArithmeticViolation as e:
        if pipe is not None or int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise
        print("[main]arithmetic on this input: %s -> the run starts again in split3 (three fp16 passes, fp32-class)" % e, file=log)
        out_path = args.output + ".modbam.bam"
        for pth in (out_path, out_path + ".bai", out_path + ".tmp"):
            if os.path.exists(pth):
                os.remove(pth)
        args.arithmetic = "split3"
        return _call_mods_once(args, log, None)


def _call_mods_once(args, log=sys.stderr, pipe=None):
    t0 = time.time()
    if pipe is None and not os.path.exists(args.model_file):
        raise ValueError("--model_file is not set right!")            # call_modifications.py:484-485
    if not os.path.exists(args.input):
        raise ValueError("--input_file does not exist!")              # :486-488
    _check_scope(args)
    if args.norm != "zscore" or args.no_decode:
        # the device extraction kernels implement the default (z-score of the CodecV1-decoded kinetics: extract_features.py:181-199,
        # 327-334); the other normalisations and raw codes go through the NumPy mirror of the reference's extraction and the
        # record-level BAM path, single GPU
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise ValueError("--norm %s%s are served by host extraction, which is single-GPU: run it without torch.distributed.run, "
                             "or use --norm zscore on CodecV1 codes for the sharded path" % (args.norm, " / --no_decode" if args.no_decode else ""))
        if args.extract != "host" or args.io != "python":
            print("[main]--norm %s%s: feature extraction on the host (--extract host --io python)" % (args.norm, " --no_decode" if args.no_decode else ""), file=log)
        args.extract, args.io = "host", "python"
    from collections import OrderedDict
    # the device model whose default arithmetic (split-mx after a clean synthetic probe) is still to be probed on this input (a stand-in
    # pipe of the CPU tests may bring a stand-in for it)
    probe_dm = getattr(pipe, "data_probe_model", None)
    if pipe is None and os.environ.get("CCSM_NULL_MODEL") == "2":      # diagnostics: the host side alone (tools/host_feed_probe.py)
        from .pipeline import HostNullPipe
        pipe = HostNullPipe()
    if pipe is None:
        from .models import ModelAttRNN
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            import torch
            args.device = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)    # one process per GPU
        model = ModelAttRNN(args.seq_len, args.layer_rnn, args.class_num, args.dropout_rate, args.hid_rnn, is_npass=True,
                            model_type=args.model_type, device=args.device, seed=args.tseed, max_batch=args.batch_size,
                            precision={"auto": 0, "split3": 3, "split-mx": 4, "hybrid": 5, "split-mx-d": 6}[getattr(args, "arithmetic", "auto")])
        para = _load_state_dict(args.model_file)
        try:
            model.load_state_dict(para)
        except RuntimeError:                                               # DDP checkpoints: strip "module." (:350-358)
            model.load_state_dict(OrderedDict((k[7:], v) for k, v in para.items()))
        model.cuda(args.device).eval()
        if int(os.environ.get("RANK", "0")) == 0:
            dm = model._dev
            probe = "" if dm.probe_sites <= 0 else (" (probe of %d sites against split3: split-mx max %.1e, 99.9 %% %.1e, %.3f %% beyond 1e-5; split-mx is "
                                                    "served only with max <= 1.25e-5 and max <= 3 x the 99.9th percentile over 65536 sites)") % (
                dm.probe_sites, dm.probe_error, dm.probe_q999, 100.0 * dm.probe_tail)
            print("[main]arithmetic: %s%s" % ({3: "split3 (three fp16 passes, fp32-class)", 4: "split-mx", 5: "hybrid (split-mx input part, three-pass "
                                               "recurrent part; forced)", 6: "split-mx-d (fp6 recurrent weights, per-row state scales; forced)"}.get(dm.precision, dm.precision), probe), file=log)
        # --batch_size (reference default 512) is the reference's sites per model call.  On the GPU-extraction paths a launch wants
        # >= 12288 sites to fill the chip (256 workgroups of 96 strand rows), and the calls do not depend on how sites are chunked
        # (every site's initial state is a function of the seed, its read's name and its position there), so the flag is only a lower bound there.
        chunk_sites = max(args.batch_size, 12288) if args.extract == "device" else args.batch_size
        probe_dm = model._dev if (model._dev.auto_precision and model._dev.precision == 4 and not args.no_data_probe) else None
        pipe = CallModsPipeline(model._dev, batch_size=chunk_sites, seed=args.tseed, extract=args.extract, norm=args.norm, no_decode=args.no_decode)
    holeids_e = None if args.holeids_e is None else _get_holes(args.holeids_e)          # extract_features.py:561-562
    holeids_ne = None if args.holeids_ne is None else _get_holes(args.holeids_ne)
    name_filter = holeids_e is not None or holeids_ne is not None
    align = args.mode == "align"
    out_path = args.output + ".modbam.bam"                             # :494
    cnt_w = cnt_mm = cnt_failed = cnt_sites = 0
    rm_pulse = not args.keep_pulse
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1 and not (args.io == "native" and args.extract == "device"):
        raise ValueError("multi-GPU call_mods needs --io native --extract device")
    if args.io == "native" and args.extract == "device":
        # file -> libccsm_bam -> ccsm_forward_reads_host -> libccsm_bam -> file; the next hole-batch is inflated / parsed and the
        # previous one deflated / written by two helper threads while the GPU works on the current one.
        # Multi-GPU (torch.distributed.run, one process per GPU; ccsmeth_amd/sharding.py): the input is cut into chunks
        # of --chunk_mb compressed MiB; every rank claims the next unclaimed chunk number (the reference's shared queue,
        # call_modifications.py:561-578, as a counter), locates the chunk's first record itself and inflates only its own chunks;
        # a chunk's records go into the rank's part file as one block-aligned run; initial states are keyed by (read name, position),
        # so every probability equals the single-GPU run's; at the end every rank copies its own runs to their place in the output
        # (in parallel) and the index is built from the writers' run tables - nothing is read back.
        from concurrent.futures import ThreadPoolExecutor
        from .bamnative import NativeBamReader, NativeBamWriter, index_write, stitch_copy, stitch_create, stitch_layout
        holes_batch = args.holes_batch
        if holes_batch == 50 and not args.holes_batch_exact:       # the reference's default: see --help
            holes_batch = 256
            print("[main]--holes_batch 50 (the reference's default) is run as 256 reads per unit of work; pass --holes_batch_exact to "
                  "keep 50 (the calls do not depend on it)", file=log)

        def filters(b):
            """(skip mask or None, site window or None, sites of the batch that will be called)"""
            skip, window = None, None
            if name_filter:
                skip = np.array([_skip_by_name(nm, holeids_e, holeids_ne) for nm in _batch_names(b)], bool)
            if align:
                from .bamnative import align_info
                mq, qs, qe, ident = align_info(b)
                askip, window = _align_skip_and_window(b.flag, mq, ident, qs, qe, b.length, args)
                skip = askip if skip is None else (skip | askip)
            sites = int(np.where((b.length > 0) & (~skip if skip is not None else True), b.n_sites, 0).sum())
            return skip, window, sites

        dist = queue = None
        chunk_bytes = max(1, int(args.chunk_mb * (1 << 20)))
        if world > 1:
            import torch.distributed as dist
            from . import sharding
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if not dist.is_initialized():
                dist.init_process_group("gloo", rank=rank, world_size=world)      # host-side bookkeeping only
            queue = sharding.ChunkQueue(sharding.open_board_store(world, rank), world, rank,
                                        sharding.n_chunks_of(os.path.getsize(args.input), chunk_bytes), dispatch=args.dispatch)
        if probe_dm is not None:
            # The selection rule on the ACTUAL input: the first <= 65536 called sites of the file go through split-mx and
            # through split3, and split-mx stays only if it is clean on them too.  Rank 0 probes and publishes the verdict: one arithmetic
            # for the whole run, whatever the number of ranks (the bytes written do not depend on the sharding).
            if rank == 0:
                head = []
                try:
                    with NativeBamReader(args.input, threads=min(args.threads, 4)) as prd:
                        # only batches that hold called sites are kept, and the scan is bounded (an input whose filters leave few sites -
                        # --holeids_e, a strict --mode align - must not be read and held whole): at most 16384 reads looked at
                        n_head = n_seen = 0
                        while n_head < 65536 and n_seen < 16384:
                            b = prd.next_batch(min(holes_batch, 64))
                            if b is None:
                                break
                            n_seen += b.n_reads
                            skip, _, sites = filters(b)
                            if sites > 0:
                                head.append((b, skip))
                                n_head += sites
                            else:
                                b.close()
                        if n_head > 0:
                            probe_dm.data_probe(lambda: (pipe.probs_of_native_batch(b, skip) for b, skip in head))
                        else:
                            print("[main]arithmetic on this input: no called site among its first %d reads: the probe on the input is skipped "
                                  "(the running shadow still applies)" % n_seen, file=log)
                except BaseException as e:      # noqa: BLE001 - the other ranks wait for the verdict: release them with the cause
                    if queue is not None:
                        queue.fail("%s: %s" % (type(e).__name__, e))
                    raise
                finally:
                    for b, _ in head:
                        b.close()
                if probe_dm.data_probe_sites > 0:
                    _print_data_probe(probe_dm, log)
            if queue is not None:
                verdict = queue.rendezvous("arithmetic", int(probe_dm.precision) if rank == 0 else None)[0]
                if rank != 0:
                    probe_dm.set_precision(int(verdict))
            # split-mx survived both probes: the rule keeps being applied to one chunk in --shadow_every for the rest of the file
            if int(probe_dm.precision) == 4 and getattr(args, "shadow_every", 64) > 0 and hasattr(pipe, "shadow_every"):
                pipe.shadow_every = int(args.shadow_every)
                pipe.shadow_limit = float(os.environ.get("CCSM_CALLMODS_SHADOW_LIMIT", "1.25e-5"))
        part_path = out_path if world == 1 else "%s.part%d" % (out_path, rank)
        runs, chunk_log = [], []          # [(chunk, file start, file end, IndexRun)], [(chunk, first voffset, end voffset, records)]
        t_work = time.time()
        try:
            with NativeBamReader(args.input, threads=args.threads) as rd:
                header = add_pg_line(rd.header_text, REF_VERSION, " ".join(sys.argv))
                if not args.no_sort and rd.n_ref == 0:
                    header = _set_coordinate_order(header)             # no reference: every record sorts equal, input order is kept
                header_inflated = rd.inflated_bytes
                first_voffset = rd.tell()
                # where the hand-over chain has to end (block headers only): rank 0 looks, the others are told
                eof_voffset = None if queue is None else queue.rendezvous("eof_voffset", rd.eof_voffset() if rank == 0 else None)[0]
                n_ref = rd.n_ref
                with NativeBamWriter(part_path, header, rd.raw_refs, rd.n_ref, threads=args.threads) as wr, \
                        ThreadPoolExecutor(1) as rpool, ThreadPoolExecutor(1) as wpool:
                    header_end = wr.flush()
                    wr.track_index(True)
                    cur = {}

                    def fetch():
                        """("batch", batch, chunk) | ("end", chunk, first voffset, end voffset, records) | None when this rank is done"""
                        if queue is None:
                            b = rd.next_batch(holes_batch)
                            return None if b is None else ("batch", b, 0)
                        while True:
                            if not cur:
                                k = queue.claim()
                                if k is None:
                                    return None
                                v0 = rd.seek_chunk(k * chunk_bytes, (k + 1) * chunk_bytes)
                                if v0 == 0:
                                    chunk_log.append((k, 0, 0, 0))
                                    continue
                                cur.update(k=k, v0=v0, n=0)
                            b = rd.next_batch(holes_batch)
                            if b is None:
                                item = ("end", cur["k"], cur["v0"], rd.tell(), cur["n"])
                                cur.clear()
                                return item
                            cur["n"] += b.n_reads
                            return ("batch", b, cur["k"])
                    nxt = rpool.submit(fetch)

                    def write(b, first, locs, prob1, tagged):
                        n = wr.write_batch(b, first, locs, prob1, tagged, rm_pulse)
                        b.close()
                        return n

                    def end_run(k):
                        a = runs[-1][2] if runs else header_end
                        e = wr.flush()
                        runs.append((k, a, e, wr.take_index()))
                        return 0
                    streaming = hasattr(pipe, "feed_native_batch")
                    todo = []                          # what still has to go to the writer, in order: ("batch", job | result, b, window) | ("end", chunk)
                    state = dict(pending=None)

                    def to_writer(fn, *a):
                        nonlocal cnt_mm
                        if state["pending"] is not None:
                            cnt_mm += state["pending"].result()
                        state["pending"] = wpool.submit(fn, *a)

                    def drain_todo():
                        nonlocal cnt_w, cnt_sites, cnt_failed
                        while todo:
                            act = todo.pop(0)
                            if act[0] == "end":
                                to_writer(end_run, act[1])
                                continue
                            _, res, b, window = act
                            first, locs, prob1, tagged, failed = pipe.finish_native_batch(res) if streaming else res
                            if window is not None:
                                first, locs, prob1, tagged = _filter_sites_by_window(first, locs, prob1, tagged, window)
                            to_writer(write, b, first, locs, prob1, tagged)
                            cnt_w += b.n_reads
                            cnt_sites += len(locs)
                            cnt_failed += failed
                    while True:
                        item = nxt.result()
                        if item is None:
                            break
                        nxt = rpool.submit(fetch)
                        if item[0] == "end":
                            _, k, v0, v1, n = item
                            chunk_log.append((k, v0, v1, n))
                            todo.append(("end", k))        # closes the run behind the chunk's last batch, whenever that one is written
                            continue
                        b = item[1]
                        skip, window, _ = filters(b)
                        if streaming:
                            # the batch's launches are queued behind the previous batch's last ones - the GPU does not drain between
                            # hole-batches, nor between the chunks of the input - and what was queued before it is complete by then
                            job = pipe.feed_native_batch(b, skip)
                            drain_todo()
                            todo.append(("batch", job, b, window))
                        else:
                            todo.append(("batch", pipe.run_native_batch(b, skip), b, window))
                            drain_todo()
                    drain_todo()
                    if state["pending"] is not None:
                        cnt_mm += state["pending"].result()
                    if world == 1:
                        end_run(0)
                work_inflated = rd.inflated_bytes - header_inflated
        except BaseException as e:      # noqa: BLE001 - the other ranks must not wait for this one's share
            if queue is not None:
                queue.fail("%s: %s%s" % (type(e).__name__, e, " - run again with --arithmetic split3" if type(e).__name__ == "ArithmeticViolation" else ""))
            if type(e).__name__ == "ArithmeticViolation":
                pipe.close(discard=True)
                if os.path.exists(part_path):
                    os.remove(part_path)
            raise
        pipe.close()
        if getattr(pipe, "shadow_chunks", 0) and rank == 0:
            print("[main]arithmetic on this input: %d chunks (%d sites) shadowed in split3 behind the probes: max |dprob| %.1e (limit %.1e) -> split-mx kept"
                  % (pipe.shadow_chunks, pipe.shadow_sites, pipe.shadow_max, pipe.shadow_limit), file=log)
        t_work = time.time() - t_work
        stats = dict(reads=cnt_w, tagged=cnt_mm, failed=cnt_failed, sites=cnt_sites, output=out_path, inflated_bytes=work_inflated, chunks=len(runs),
                     seconds_work=t_work, ranks_seen=1, distinct_devices=1)
        ordered = [(k, part_path, a, e, ir) for k, a, e, ir in runs]
        t_stitch = time.time()
        if world > 1:
            # gather + the two barriers around the stitch go through the queue's store with its error key polled (ChunkQueue.rendezvous):
            # a rank that fails anywhere from here on releases the others with its error instead of leaving them in a collective
            try:
                gathered = queue.rendezvous("gather", dict(rank=rank, header_end=header_end, runs=runs, chunks=chunk_log,
                                                           counts=(cnt_w, cnt_mm, cnt_failed, cnt_sites), inflated=work_inflated, work=t_work,
                                                           device=sharding.device_identity()))
                cnt_w, cnt_mm, cnt_failed, cnt_sites = (sum(g["counts"][k] for g in gathered) for k in range(4))
                n_chain = sharding.verify_chain(first_voffset, [c for g in gathered for c in g["chunks"]], n_chunks=queue.n_chunks, eof_voffset=eof_voffset)
                if n_chain != cnt_w:
                    raise RuntimeError("the chunks hold %d records but %d were written" % (n_chain, cnt_w))
                stats.update(reads=cnt_w, tagged=cnt_mm, failed=cnt_failed, sites=cnt_sites, rank_inflated_bytes=[g["inflated"] for g in gathered],
                             rank_chunks=[len(g["runs"]) for g in gathered], rank_seconds_work=[g["work"] for g in gathered])
                census = sharding.device_census([g.get("device") for g in gathered], world)
                stats.update(ranks_seen=census["ranks_seen"], distinct_devices=census["distinct_devices"], collective_library=sharding.collective_library())
                ordered = sorted((k, "%s.part%d" % (out_path, g["rank"]), a, e, ir) for g in gathered for k, a, e, ir in g["runs"])
                spans = [(p, a, e) for _, p, a, e, _ in ordered]
                dst, total = stitch_layout(header_end, spans)
                if rank == 0:
                    stitch_create(out_path, part_path, header_end, total)
                queue.rendezvous("stitch_created")
                mine = [i for i, sp in enumerate(spans) if sp[0] == part_path]
                stitch_copy(out_path, [spans[i] for i in mine], [dst[i] for i in mine])      # every rank moves its own runs, at the same time
                queue.rendezvous("stitch_copied")
                os.remove(part_path)
            except BaseException as e:      # noqa: BLE001
                queue.fail("%s: %s" % (type(e).__name__, e))      # first failure wins: an echo of another rank's error does not replace it
                for stale in (part_path, out_path if rank == 0 else None):      # no half-stitched output, no part files left behind
                    try:
                        if stale and os.path.exists(stale):
                            os.remove(stale)
                    except OSError:
                        pass
                raise
            shifts = [d - a for d, (_, a, _) in zip(dst, spans)]
        else:
            shifts = [0] * len(ordered)
        t_stitch = time.time() - t_stitch
        try:
            if rank == 0:
                t_idx = time.time()
                indexed = False
                if not args.no_sort:
                    # the reference's samtools sort + index (call_modifications.py:592-607): records that are already in coordinate order
                    # (every unaligned HiFi BAM; a sorted aligned one) are indexed from the writers' run tables without reading the file
                    # back; only an unsorted input takes the real sort
                    try:
                        indexed, _ = index_write(out_path + ".bai", n_ref, [ir for *_, ir in ordered], shifts)
                    except IOError as e:
                        print("[post_process] WARNING: writing the index from the run tables failed (%s); falling back to a pass over the file" % e, file=log)
                    if indexed:
                        print("[post_process] bam_sort_index costs %.2f seconds (already in order: indexed from the writers' run tables)" % (time.time() - t_idx),
                              file=log)
                    else:
                        _post_sort_index(out_path, args, log)
                stats.update(seconds_stitch=t_stitch, seconds_index=time.time() - t_idx)
                print("wrote {} reads, in which {} were added mm tags".format(cnt_w, cnt_mm), file=log)     # :456
                print("[main]call_mods costs %.1f seconds.. (%d reads skipped/failed; %d rank(s) on %d distinct GPU(s); ccsmeth_amd %s)" %
                      (time.time() - t0, cnt_failed, world, stats.get("distinct_devices", 1), __version__), file=log)
                if os.environ.get("CCSM_CALLMODS_REPORT"):      # machine-readable run summary (tools/host_feed_probe.py)
                    import json
                    with open(os.environ["CCSM_CALLMODS_REPORT"], "w") as rf:
                        json.dump(dict({k: v for k, v in stats.items() if k != "output"}, seconds=time.time() - t0, world=world), rf)
        except BaseException as e:      # noqa: BLE001 - the other ranks wait for rank 0's index at "done"
            if queue is not None:
                queue.fail("%s: %s" % (type(e).__name__, e))
            raise
        if world > 1:
            queue.rendezvous("done")        # (error-aware: rank 0's index phase may have failed)
            dist.barrier()                  # every rank has passed "done": rank 0, which hosts the store, may leave now
        return stats
    with BamReader(args.input) as rd:
        header = add_pg_line(rd.header_text, REF_VERSION, " ".join(sys.argv))
        if not args.no_sort and not rd.references:
            header = _set_coordinate_order(header)
        with BamWriter(out_path, header, rd.references) as wr:
            batch = []

            probe_seen = 0

            def flush():
                nonlocal cnt_w, cnt_mm, cnt_failed
                if not batch:
                    return
                rds = [_read_of(r) for r in batch]
                if name_filter:       # a filtered read goes through as one without usable kinetics
                    rds = [r._replace(fi=np.empty(0, np.uint8)) if _skip_by_name(r.name, holeids_e, holeids_ne) else r for r in rds]
                windows = None
                if align:
                    L = np.array([len(r.seq) for r in batch], np.int64)
                    info = [_cigar_align_info(r.cigar, len(r.seq)) for r in batch]
                    qs, qe, ident = (np.array([x[k] for x in info]) for k in range(3))
                    askip, windows = _align_skip_and_window(np.array([r.flag for r in batch]), np.array([r.mapq for r in batch]), ident, qs, qe, L, args)
                    rds = [r._replace(fi=np.empty(0, np.uint8)) if sk else r for r, sk in zip(rds, askip)]
                nonlocal probe_dm, probe_seen
                if probe_dm is not None:
                    # the selection rule on this input's first reads (the native path probes up to 65536 sites; here: hole-batch by hole-batch
                    # until 8192 sites have been compared - a first batch without sites decides nothing).  RAW probabilities, as in the
                    # native path and in the library's rule: the calls' own values are normalised and rounded to 6 decimals
                    def p2():
                        pipe.raw_log = []
                        pipe.run(rds)
                        raw, pipe.raw_log = pipe.raw_log, None
                        yield np.concatenate(raw + [np.empty((0, 2), np.float32)])
                    probe_dm.data_probe(p2)
                    probe_seen += probe_dm.data_probe_sites
                    if probe_dm.data_probe_sites > 0:
                        _print_data_probe(probe_dm, log)
                    if probe_seen >= 8192 or int(probe_dm.precision) != 4:
                        probe_dm = None
                calls, failed = pipe.run(rds)
                cnt_failed += failed
                for ri, (rec, c) in enumerate(zip(batch, calls)):
                    if windows is not None and c.mm_flag:
                        keep = (np.asarray(c.locs) >= windows[0][ri]) & (np.asarray(c.locs) < windows[1][ri])
                        if not keep.all():
                            locs_k = [int(x) for x in np.asarray(c.locs)[keep]]
                            probs_k = np.asarray(c.probs)[keep]
                            if locs_k:
                                c = c._replace(locs=locs_k, probs=probs_k, mm=_convert_locs_to_mmtag(locs_k, rds[ri].seq),
                                               ml=_convert_probs_to_mltag(list(probs_k)))
                            else:
                                c = c._replace(mm_flag=0)
                    old = [(t, v) for t, _, v in rec.tags]
                    mm_vals = c.mm if c.mm_flag else None
                    kept = {t for t, _ in _refill_tags(old, None, None, rm_pulse)}
                    rec.tags = [tv for tv in rec.tags if tv[0] in kept]
                    if mm_vals is not None:
                        rec.tags.append(("MM", "Z", "C+m?," + ",".join(map(str, mm_vals)) + ";"))
                        rec.tags.append(("ML", "BC", np.asarray(c.ml, np.uint8)))
                    wr.write(rec)
                    cnt_w += 1
                    cnt_mm += c.mm_flag
                batch.clear()

            for rec in rd:
                batch.append(rec)
                if len(batch) == args.holes_batch:
                    flush()
            flush()
    pipe.close()
    _post_sort_index(out_path, args, log)
    print("wrote {} reads, in which {} were added mm tags".format(cnt_w, cnt_mm), file=log)     # :456
    print("[main]call_mods costs %.1f seconds.. (%d reads skipped/failed; ccsmeth_amd %s)" %
          (time.time() - t0, cnt_failed, __version__), file=log)
    return dict(reads=cnt_w, tagged=cnt_mm, failed=cnt_failed, output=out_path)


def main(argv=None):
    args = build_parser().parse_args(argv)
    return call_mods(args)


if __name__ == "__main__":
    main()
