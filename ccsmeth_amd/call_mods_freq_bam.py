"""Aggregate-mode methylation frequency — host mirror of reference ccsmeth/call_mods_freq_bam.py:102-107 (_cal_mod_prob),
:221-237 (_get_normalized_histo), :265-305 (_cal_modfreq_in_aggregate_mode), with the model on libccsm's HIP kernel.

`AggrModel` replaces AggrAttRNN + load_state_dict (call_mods_freq_bam.py:316-342); windows are built on the device from
the per-site histogram table, and the per-region seeded torch.randn stream is reproduced by the library, so results match
the (deterministic) reference.  No CPU fallback."""
import ctypes as C

import numpy as np

from . import _lib

_ML2PROB = np.array([round(m / 256.0 + 0.000001, 6) if m > 0 else 0.0 for m in range(256)], dtype=np.float64)


def _cal_mod_prob(ml_value):
    """ML byte -> probability: round(ml/256 + 1e-6, 6), 0 for ml == 0 (call_mods_freq_bam.py:102-107)."""
    return round(ml_value / float(256) + 0.000001, 6) if ml_value > 0 else 0


def _get_normalized_histo(probs, cov_cf=4, binsize=20):
    """20-bin histogram over [0,1] / its L2 norm, rounded to 6 dp (call_mods_freq_bam.py:221-237)."""
    assert len(probs) >= cov_cf
    hist = np.histogram(probs, bins=binsize, range=[0, 1])[0]
    return np.round(hist / np.linalg.norm(hist), 6)


def histos_from_ml(ml_arrays, cov_cf=4, binsize=20):
    """Vectorised form of the two functions above for many sites: list of uint8 ML arrays -> (keep mask, (n_keep, 20)
    float64 normalised histograms, coverages).  Bit-identical to np.histogram on the rounded probabilities: the bin of a
    value is computed exactly as np.histogram does for uniform bins (floor(p * 20), last edge inclusive)."""
    keep, hists, covs = [], [], []
    for ml in ml_arrays:
        ml = np.asarray(ml, dtype=np.uint8)
        if len(ml) >= cov_cf:
            keep.append(True)
            hists.append(_get_normalized_histo(_ML2PROB[ml], cov_cf, binsize))
            covs.append(len(ml))
        else:
            keep.append(False)
    return np.array(keep, bool), (np.array(hists) if hists else np.zeros((0, binsize))), np.array(covs, np.int64)


class AggrModel:
    """Device model for attbigru_b11 (seq_len 11, hidden 32, bin_size 20).  state_dict keys as in the reference checkpoint
    (a leading 'module.' is stripped like call_mods_freq_bam.py:329-337)."""

    def __init__(self, state_dict, device=0, tseed=1234, stream_sites=1 << 20, seq_len=11, hid_rnn=32, bin_size=20,
                 model_type="attbigru"):
        if model_type != "attbigru":
            raise ValueError("--model_type not right!")          # call_mods_freq_bam.py:323
        if (seq_len, hid_rnn, bin_size) != (11, 32, 20):
            raise ValueError("this build implements attbigru_b11: seq_len 11, hid_rnn 32, bin_size 20")
        sd = {}
        for k, v in state_dict.items():
            v = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            sd[k[7:] if k.startswith("module.") else k] = np.ascontiguousarray(v, dtype=np.float32)
        self._keep = sd
        w = _lib.AggrWeights()
        p = lambda k: sd[k].ctypes.data  # noqa: E731
        for d, sfx in enumerate(("", "_reverse")):
            w.weight_ih[d], w.weight_hh[d] = p("rnn.weight_ih_l0" + sfx), p("rnn.weight_hh_l0" + sfx)
            w.bias_ih[d], w.bias_hh[d] = p("rnn.bias_ih_l0" + sfx), p("rnn.bias_hh_l0" + sfx)
        w.att_wa, w.att_ua, w.att_va = p("_att3.Wa.weight"), p("_att3.Ua.weight"), p("_att3.va.weight")
        w.fc1_weight, w.fc1_bias = p("fc1.weight"), p("fc1.bias")
        self._lib = _lib.load()
        self.handle = C.c_void_p()
        _lib.check(self._lib.ccsm_aggr_create(C.byref(w), int(device), int(tseed), int(stream_sites), C.byref(self.handle)))
        self.device = int(device)
        self.stream_pos = 0

    def new_region(self):
        """The reference re-seeds at the start of every region (call_mods_freq_bam.py:313)."""
        self.stream_pos = 0

    def close(self):
        if self.handle:
            self._lib.ccsm_aggr_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward_raw(self, refposes, histos, only_close=False):
        """fc1 outputs (M,) float32 for the sites of one call; advances the region's random stream by 64 values per site."""
        if bool(only_close) != getattr(self, "_only_close", False):
            _lib.check(self._lib.ccsm_aggr_set_only_close(self.handle, int(bool(only_close))))
            self._only_close = bool(only_close)
        pos = np.ascontiguousarray(refposes, dtype=np.int64)
        h = np.ascontiguousarray(histos, dtype=np.float32)
        m = len(pos)
        assert h.shape == (m, 20)
        out = np.empty(m, np.float32)
        _lib.check(self._lib.ccsm_aggr_forward_host(self.handle, m, pos.ctypes.data, h.ctypes.data, self.stream_pos,
                                                    out.ctypes.data, None))
        self.stream_pos += 64 * m
        return out


def _cal_modfreq_in_aggregate_mode(refposes, refposes_histos, model, seq_len=11, only_close=False):
    """call_mods_freq_bam.py:265-305: per-site probabilities round(clip(y, 0, 1), 6) (float32), or None for no sites."""
    if len(refposes) == 0:
        return None
    y = model.forward_raw(refposes, np.stack(refposes_histos), only_close)
    return list(np.round(np.clip(y, 0, 1), 6))


# ---------------------------------------------------------------------------------------------------------------------------
# `call_freqb`: aligned modbam -> per-site modification frequency (count / aggregate mode), the caller side of AggrModel.
# Host mirror of call_mods_freq_bam.py:51-85 (regions), :209-230 (count mode), :244-262 (discretize), :308-437 (one region),
# :455-585 (projection on the reference, strand combining, motif filter), :611-668 (writer).  The per-record work (filters,
# MM/ML, CIGAR walk) is libccsm_bam's ccsm_bam_modcalls_of_batch; no pysam, no BAM index: the file is streamed once and the
# calls are grouped by region afterwards, which gives the same per-region lists as the reference's fetch() per region.
# ---------------------------------------------------------------------------------------------------------------------------
import math
import os
import sys
import time

_IUPAC = {'A': 'A', 'T': 'T', 'C': 'C', 'G': 'G', 'R': 'AG', 'M': 'AC', 'S': 'CG', 'Y': 'CT', 'K': 'GT', 'W': 'AT', 'B': 'CGT',
          'D': 'AGT', 'H': 'ACT', 'V': 'ACG', 'N': 'ACGT'}
_PAIR = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A', 'N': 'N', 'W': 'W', 'S': 'S', 'M': 'K', 'K': 'M', 'R': 'Y', 'Y': 'R', 'B': 'V',
         'V': 'B', 'D': 'H', 'H': 'D', 'Z': 'Z'}


def get_motif_seqs(motifs):
    """process_utils.py:140-170: comma-separated IUPAC motifs -> list of plain sequences (order as the reference builds it)."""
    out = []
    for m in motifs.strip().split(','):
        seqs = ['']
        for b in m.strip().upper():
            seqs = [s + x for s in seqs for x in _IUPAC[b]]
        out += seqs
    return out


def complement_seq(seq):
    """process_utils.py:106-118: reverse complement, unknown letters -> N."""
    return ''.join(_PAIR.get(x, 'N') for x in seq[::-1])


def read_fasta(path):
    """DNAReference (utils/ref_reader.py:34-52): contig name = header up to the first blank, sequence upper-cased."""
    names, contigs = [], {}
    name, parts = None, []
    with open(path, "r") as rf:
        for line in rf:
            if line.startswith('>'):
                if name is not None:
                    contigs[name] = ''.join(parts)
                    if name not in names:
                        names.append(name)
                name, parts = line.strip()[1:].split(' ')[0], []
            else:
                parts.append(line.strip().upper())
    if name is not None:
        contigs[name] = ''.join(parts)
        if name not in names:
            names.append(name)
    return names, contigs


def _get_reference_chunks(dnacontigs, contig_str, chunk_len=300000, motifs="CG"):
    """call_mods_freq_bam.py:51-85: (contig, start, end) in sorted contig order; a CG across a boundary goes to the left chunk."""
    if contig_str is not None:
        if os.path.isfile(contig_str):
            with open(contig_str, "r") as rf:
                contigs = sorted(set(rf.read().splitlines()))
        else:
            contigs = sorted(set(contig_str.strip().split(",")))
    else:
        contigs = sorted(dnacontigs.keys())
    chunks = []
    for contig in contigs:
        if contig not in dnacontigs:
            raise ValueError("contig {} is not in --ref".format(contig))
        n = len(dnacontigs[contig])
        for i in range(0, n, chunk_len):
            chunks.append((contig, i, i + chunk_len if i + chunk_len < n else n))
    if motifs == "CG":
        for idx in range(1, len(chunks)):
            pre_ref, pre_s, pre_e = chunks[idx - 1]
            cur_ref, cur_s, cur_e = chunks[idx]
            if pre_ref != cur_ref:
                continue
            if dnacontigs[pre_ref][(pre_e - 1):(pre_e + 1)] == "CG":
                chunks[idx - 1] = (pre_ref, pre_s, pre_e + 1)
                chunks[idx] = (cur_ref, cur_s + 1, cur_e)
    return chunks


def _motif_site_mask(seq, regions, motifs, mod_loc):
    """uint8 per base: bit 0 = --refsites_all site of the forward strand, bit 1 = of the reverse strand, found exactly as
    call_mods_freq_bam.py:473-479 does: inside each region's own slice (a motif cut by a region boundary is not a site)."""
    s = np.frombuffer(seq.encode("ascii"), dtype=np.uint8)
    n = len(s)
    mask = np.zeros(n, np.uint8)
    if n == 0:
        return mask
    mlen = len(motifs[0])
    starts = np.array([r[0] for r in regions], np.int64)
    ends = np.array([r[1] for r in regions], np.int64)
    for motif in set(motifs):
        for bit, m, site_off in ((1, motif, mod_loc), (2, complement_seq(motif), mlen - 1 - mod_loc)):
            if n < mlen:
                continue
            hit = np.ones(n - mlen + 1, bool)
            for k, ch in enumerate(m.encode("ascii")):
                hit &= s[k:n - mlen + 1 + k] == ch
            st = np.flatnonzero(hit)
            if len(st) == 0:
                continue
            ri = np.searchsorted(starts, st, side="right") - 1          # region of the motif's first base
            ok = (ri >= 0) & (st + mlen <= ends[np.maximum(ri, 0)])
            mask[st[ok] + site_off] |= bit
    return mask


def discretize_score(modprob, coverage):
    """A site frequency pushed towards whole reads (call_mods_freq_bam.py:244-262; the idea is pb-CpG-tools'): above 0.66 the
    expected number of modified reads is rounded up, at or below 0.33 down, in between it is kept to two decimals.
    -> (modified reads, unmodified reads, modified / (modified + unmodified))."""
    n_reads = int(coverage)
    expected = modprob * float(coverage)
    snap = math.ceil if modprob > 0.66 else math.floor if modprob <= 0.33 else None
    modified = int(snap(expected)) if snap is not None else round(coverage * modprob, 2)
    unmodified = n_reads - modified
    return modified, unmodified, (float(modified) / (modified + unmodified) if modified != 0 else 0.0)


def _write_one_line(beditem, wf, is_bed):
    """call_mods_freq_bam.py:611-620."""
    ref_name, refpos, strand, cov, met, metprob = beditem
    if is_bed:
        wf.write("\t".join([ref_name, str(refpos), str(refpos + 1), ".", str(cov), strand, str(refpos), str(refpos + 1),
                            "0,0,0", str(cov), str(int(round(metprob * 100 + 0.001, 0)))]) + "\n")
    else:
        wf.write("\t".join([ref_name, str(refpos), str(refpos + 1), strand, ".", ".", str(met), str(cov - met), str(cov),
                            str(round(metprob + 0.000001, 4)), "."]) + "\n")


class _CountTables:
    """Per ML byte: is the call kept under --prob_cf, is it a modified call (call_mods_freq_bam.py:211-216)."""

    def __init__(self, prob_cf):
        probs = [float(p) for p in _ML2PROB]
        self.kept = np.array([not (abs(p - (1 - p)) < prob_cf) for p in probs], np.int64)
        self.mod = np.array([(not (abs(p - (1 - p)) < prob_cf)) and p > 0.5 for p in probs], np.int64)
        # bin of the ML byte's probability in np.histogram(probs, bins=20, range=[0, 1])
        self.bin = np.array([int(np.argmax(np.histogram([p], bins=20, range=[0, 1])[0])) for p in probs], np.int64)


def _count_infos(n, nf, nm, no_amb_cov):
    """(coverage, modified, frequency) per site from the totals / kept / modified-and-kept counts; None where n == 0."""
    out = []
    for a, f, m in zip(n.tolist(), nf.tolist(), nm.tolist()):
        if a == 0:
            out.append(None)
            continue
        modfreq = m / float(f) if f > 0 else 0.
        if no_amb_cov:
            out.append((f, m, modfreq))
        else:
            out.append((a, np.round(a * modfreq, 2) if f != a else m, modfreq))
    return out


def _call_modfreq_of_one_region(pos, ml, hap, args, tables, model):
    """_call_modfreq_of_one_region (call_mods_freq_bam.py:423-451) on the calls of one region and strand dictionary, given
    as arrays sorted by position.  -> [(refpos, info_all, info_hp1, info_hp2)] ascending in refpos."""
    if len(pos) == 0:
        return []
    first = np.flatnonzero(np.r_[True, pos[1:] != pos[:-1]])
    refposes = pos[first].tolist()
    kept, mod = tables.kept[ml], tables.mod[ml]
    sel = [np.ones(len(pos), np.int64)]
    if args.no_hap:
        sel += [np.zeros(len(pos), np.int64)] * 2
    else:
        sel += [(hap == 1).astype(np.int64), (hap == 2).astype(np.int64)]
    red = lambda v: np.add.reduceat(v, first)  # noqa: E731
    cols = []                                   # per all / hp1 / hp2: list of infos
    for s in sel:
        n, nf, nm = red(s), red(s * kept), red(s * mod)
        if args.call_mode == "count":
            cols.append(_count_infos(n, nf, nm, args.no_amb_cov))
            continue
        infos = _count_infos(np.where(n < args.cov_cf, n, 0), nf, nm, args.no_amb_cov)   # low coverage: count mode
        cols.append((infos, n, s))
    if args.call_mode == "count":
        return list(zip(refposes, cols[0], cols[1], cols[2]))
    if args.call_mode != "aggregate":
        raise ValueError("wrong --call_mode")
    # aggregate mode (:308-420): torch.manual_seed(tseed) once per call of this function, then all -> hp1 -> hp2
    model.new_region()
    site_of_row = np.repeat(np.arange(len(first)), np.diff(np.r_[first, len(pos)]))
    out_cols = []
    for infos, n, s in cols:
        high = np.flatnonzero(n >= args.cov_cf)
        if len(high):
            rows = np.flatnonzero(s)
            hist = np.bincount(site_of_row[rows] * 20 + tables.bin[ml[rows]], minlength=len(first) * 20).reshape(-1, 20)[high]
            histos = np.round(hist / np.sqrt((hist * hist).sum(1, dtype=np.float64))[:, None], 6)
            probs = _cal_modfreq_in_aggregate_mode([refposes[i] for i in high], list(histos), model, args.seq_len, args.only_close)
            for i, p in zip(high.tolist(), probs):
                cov = int(n[i])
                if args.discrete:
                    d_cnt_mod, _, d_modprob = discretize_score(p, cov)
                    infos[i] = (cov, d_cnt_mod, d_modprob)
                else:
                    infos[i] = (cov, round(cov * p, 2), p)
        out_cols.append(infos)
    return list(zip(refposes, out_cols[0], out_cols[1], out_cols[2]))


def _bam_ref_names(raw_refs, n_ref):
    """Names of the binary reference list of a BAM header (l_name, name\\0, l_ref per entry)."""
    names, o = [], 0
    for _ in range(n_ref):
        ln = int.from_bytes(raw_refs[o:o + 4], "little")
        names.append(raw_refs[o + 4:o + 4 + ln - 1].decode("ascii", "replace"))
        o += 8 + ln
    return names


def _bed_of_contig(name, seq, regions, rows, motifs_filter, args, tables, model):
    """_readmods_to_bed_of_one_region (call_mods_freq_bam.py:538-585) for every region of one contig.
    rows = (pos, strand, ml, hap) of the contig's calls.  -> (bed_all, bed_hp1, bed_hp2)."""
    pos, strand, ml, hap = rows
    beds = ([], [], [])
    if len(pos) == 0:
        return beds
    starts = np.array([r[1] for r in regions], np.int64)
    ends = np.array([r[2] for r in regions], np.int64)
    pos = pos.astype(np.int64)
    ridx = np.searchsorted(starts, pos, side="right") - 1
    ok = (ridx >= 0) & (pos < ends[np.maximum(ridx, 0)])
    comb = args.motifs == "CG" and not args.no_comb
    if comb:                                    # :538-548: a reverse-strand call at the G counts for the C one base to the left
        ok &= ~((strand == 1) & (pos == 0))
        pos = np.where(strand == 1, pos - 1, pos)
        strand = np.zeros_like(strand)
    pos, strand, ml, hap, ridx = pos[ok], strand[ok], ml[ok], hap[ok], ridx[ok]
    order = np.lexsort((pos, strand, ridx))     # region, then forward dictionary before reverse, then position
    pos, strand, ml, hap, ridx = pos[order], strand[order], ml[order], hap[order], ridx[order]
    key = ridx * 2 + strand
    cut = np.flatnonzero(np.r_[True, key[1:] != key[:-1], True])
    mlen = len(motifs_filter[0]) if motifs_filter is not None else 0
    mset = set(motifs_filter) if motifs_filter is not None else None
    for a, b in zip(cut[:-1].tolist(), cut[1:].tolist()):
        rev = bool(strand[a])
        res = _call_modfreq_of_one_region(pos[a:b], ml[a:b], hap[a:b], args, tables, model)
        sign = "-" if rev else "+"
        for refpos, total_info, hp1_info, hp2_info in res:
            if mset is not None:
                if rev:
                    s0, s1 = refpos - (mlen - 1 - args.mod_loc), refpos + args.mod_loc + 1
                    motif_seq = complement_seq(seq[s0:s1])
                else:
                    s0, s1 = refpos - args.mod_loc, refpos + mlen - args.mod_loc
                    motif_seq = seq[s0:s1]
                if motif_seq not in mset:
                    continue
            for bed, info in zip(beds, (total_info, hp1_info, hp2_info)):
                if info is not None:
                    bed.append((name, refpos, sign, info[0], info[1], info[2]))
    return beds


def call_mods_frequency_from_bamfile(args, log=sys.stderr, model=None):
    """call_mods_freq_bam.py:671-735.  `model` (tests): an object with new_region() / forward_raw() instead of AggrModel."""
    from . import bamnative
    t0 = time.time()
    if args.call_mode == "aggregate" and model is None and not (args.aggre_model and os.path.exists(args.aggre_model)):
        raise ValueError("--aggre_model is not set right!")
    if not args.input_bam.endswith(".bam"):
        raise ValueError("--input_bam not a bam file!")
    if not os.path.exists(args.input_bam):
        raise ValueError("--input_bam does not exist!")
    if not os.path.exists(args.ref):
        raise ValueError("--ref does not exist!")
    if args.call_mode == "aggregate":
        if args.model_type != "attbigru":
            raise ValueError("--model_type not right!")
        if (args.seq_len, args.layer_rnn, args.hid_rnn, args.bin_size, args.class_num) != (11, 1, 32, 20, 1):
            raise ValueError("this build implements the aggregate model attbigru_b11: --seq_len 11 --layer_rnn 1 --hid_rnn 32 "
                             "--bin_size 20 --class_num 1")
    out_dir = os.path.dirname(os.path.abspath(args.output))
    os.makedirs(out_dir, exist_ok=True)

    _, dnacontigs = read_fasta(args.ref)
    motifs = get_motif_seqs(args.motifs)
    motifs_filter = motifs if (args.refsites_only or args.refsites_all) else None
    chunks = _get_reference_chunks(dnacontigs, args.contigs, args.chunk_len, args.motifs)
    regions_of = {}
    for c in chunks:
        regions_of.setdefault(c[0], []).append(c)

    if args.call_mode == "aggregate" and model is None:
        import torch
        model = AggrModel(torch.load(args.aggre_model, map_location="cpu"), device=getattr(args, "device", 0), tseed=args.tseed)
    tables = _CountTables(args.prob_cf)

    # The input is coordinate-sorted (the reference fetches its regions through the BAM index, which needs that order), so a contig is
    # complete as soon as a record of a later one shows up: its rows are turned into bed lines right then and dropped, and the lines
    # go to per-contig spool files that are concatenated in contig-name order at the end (the order of the reference's region list).
    # Host memory is bounded by the largest contig instead of the whole file.
    import shutil
    import tempfile
    rows_of = {}
    done = set()
    n_rec = n_used = n_sites = 0
    fext = "bed" if args.bed else "freq.txt"
    paths = [args.output + ".{}.{}.{}".format(args.call_mode, w, fext) for w in ("all", "hp1", "hp2")]
    spool_dir = tempfile.mkdtemp(prefix="ccsm_freqb_", dir=out_dir)
    spooled = {}
    current = None                       # reference id of the run of records being collected

    def finish(t, names):
        nonlocal n_sites
        done.add(t)
        parts = rows_of.pop(t, None)
        if parts is None or not (0 <= t < len(names)) or names[t] not in regions_of:
            return
        name = names[t]
        rows = tuple(np.concatenate([p[k] for p in parts]) for k in range(4))
        beds = _bed_of_contig(name, dnacontigs[name], regions_of[name], rows, motifs_filter, args, tables, model)
        n_sites += len(beds[0])
        files = [os.path.join(spool_dir, "%d.%d" % (t, k)) for k in range(3)]
        for fp, bed in zip(files, beds):
            with open(fp, "w") as wf:
                for item in bed:
                    _write_one_line(item, wf, args.bed)
        spooled[t] = files

    try:
        with bamnative.NativeBamReader(args.input_bam, threads=max(1, args.threads)) as rd:
            names = _bam_ref_names(rd.raw_refs, rd.n_ref)
            if rd.n_ref == 0:
                raise ValueError("file has no sequences defined - please make sure that the reads are aligned to the genome reference!")
            masks = None
            if args.refsites_all:
                masks = [(_motif_site_mask(dnacontigs[nm], [(r[1], r[2]) for r in regions_of[nm]], motifs, args.mod_loc)
                          if nm in regions_of else None) for nm in names]
            while True:
                batch = rd.next_batch(4096)
                if batch is None:
                    break
                tid, pos, strand, ml, hap, seen, used = bamnative.modcalls_of_batch(
                    batch, mapq=args.mapq, identity=args.identity, no_supplementary=args.no_supplementary, base_clip=args.base_clip,
                    refsites_all=args.refsites_all, hap_tag=args.hap_tag, site_masks=masks, threads=max(1, args.threads))
                batch.close()
                n_rec += seen
                n_used += used
                if len(tid):
                    # the rows come in record order: runs of equal reference id.  A contig whose run has ended must not come back
                    # (its bed lines are already written): checked row by row, so the verdict does not depend on where the batches
                    # of 4096 records happen to be cut
                    cut = np.flatnonzero(np.diff(tid)) + 1
                    starts = np.concatenate(([0], cut))
                    ends = np.concatenate((cut, [len(tid)]))
                    for a, e in zip(starts.tolist(), ends.tolist()):
                        t = int(tid[a])
                        if current is not None and t != current:
                            finish(current, names)
                        if t in done:
                            raise ValueError("--input_bam is not coordinate-sorted (records of %s after those of a later contig): sort and index "
                                             "it first (samtools sort / call_mods without --no_sort), as the reference's region fetches require"
                                             % (names[t] if 0 <= t < len(names) else t))
                        current = t
                        if 0 <= t < len(names) and names[t] in regions_of:
                            rows_of.setdefault(t, []).append((pos[a:e], strand[a:e], ml[a:e], hap[a:e]))
            for t in sorted(rows_of):
                finish(t, names)
        files = [open(p, "w") for p in paths]
        for _, t in sorted((names[t], t) for t in spooled):      # by contig name, as before; duplicate @SQ names keep both contigs
            for wf, fp in zip(files, spooled[t]):
                with open(fp, "r") as rf:
                    shutil.copyfileobj(rf, wf)
        for wf in files:
            wf.close()
    finally:
        shutil.rmtree(spool_dir, ignore_errors=True)
    for p in paths:
        if os.path.getsize(p) == 0:
            os.remove(p)
            continue
        if args.sort or args.gzip:
            _sort_bed_file(p)
        if args.gzip:
            _bgzip_and_tabix(p)
    log.write("[call_freqb] {} records ({} used), {} sites, {:.1f} s\n".format(n_rec, n_used, n_sites, time.time() - t0))
    return n_sites


def _sort_bed_file(path):
    """--sort (call_mods_freq_bam.py:655-658, bedtools sort): by contig name, then start; ties keep their order."""
    with open(path, "r") as rf:
        lines = rf.readlines()
    lines.sort(key=lambda ln: (ln.split("\t", 2)[0], int(ln.split("\t", 2)[1])))
    with open(path, "w") as wf:
        wf.writelines(lines)


def _reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def _reg2bins(beg, end):
    """Bins that may hold records overlapping [beg, end) (SAM spec 5.3)."""
    end -= 1
    out = [0]
    for shift, base in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
        out += list(range(base + (beg >> shift), base + (end >> shift) + 1))
    return out


def _bgzip_and_tabix(path):
    """--gzip (call_mods_freq_bam.py:659-663: pysam.tabix_index(bedfile, force=True, preset="bed", keep_original=False)):
    BGZF-compress the (sorted) file to <path>.gz, write the tabix index <path>.gz.tbi (TBI v1: UCSC/BED preset = sequence, begin,
    end in columns 1-3, 0-based half-open; binning + 16 kb linear index over BGZF virtual offsets) and remove the original."""
    import struct
    from .bamio import bgzf_compress_block
    eof = bytes([0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    with open(path, "rb") as rf:
        data = rf.read()
    blk = 0xff00
    starts, coffs = [], []
    with open(path + ".gz", "wb") as wf:
        for i in range(0, len(data), blk):
            starts.append(i)
            coffs.append(wf.tell())
            wf.write(bgzf_compress_block(data[i:i + blk]))
        end_coff = wf.tell()
        wf.write(eof)

    def voff(p):
        if p >= len(data):
            return end_coff << 16
        k = p // blk
        return (coffs[k] << 16) | (p - starts[k])
    names, idx = [], {}
    pos = 0
    prev = None
    for line in data.split(b"\n"):
        ln = len(line) + 1
        if line and not line.startswith(b"#"):
            f = line.split(b"\t", 3)
            name, beg, end = f[0].decode(), int(f[1]), int(f[2])
            if name not in idx:
                if prev is not None and name in names:
                    raise ValueError("the file is not sorted by sequence name")
                names.append(name)
                idx[name] = dict(bins={}, lin=[], n=0, beg=voff(pos), end=0)
            elif name != prev:
                raise ValueError("the file is not sorted by sequence name")
            d = idx[name]
            v0, v1 = voff(pos), voff(pos + ln)
            end = max(end, beg + 1)
            chunks = d["bins"].setdefault(_reg2bin(beg, end), [])
            if chunks and chunks[-1][1] == v0:
                chunks[-1][1] = v1                                # adjacent records of a bin form one chunk
            else:
                chunks.append([v0, v1])
            w1 = (end - 1) >> 14
            if len(d["lin"]) <= w1:
                d["lin"] += [None] * (w1 + 1 - len(d["lin"]))
            for w in range(beg >> 14, w1 + 1):
                if d["lin"][w] is None:
                    d["lin"][w] = v0
            d["n"] += 1
            d["end"] = v1
            prev = name
        pos += ln
    out = bytearray(b"TBI\x01")
    nm = b"".join(n.encode() + b"\x00" for n in names)
    out += struct.pack("<8i", len(names), 0x10000, 1, 2, 3, ord("#"), 0, len(nm)) + nm
    for n in names:
        d = idx[n]
        out += struct.pack("<i", len(d["bins"]) + 1)
        for b_, chunks in sorted(d["bins"].items()):
            out += struct.pack("<Ii", b_, len(chunks))
            for c0, c1 in chunks:
                out += struct.pack("<QQ", c0, c1)
        out += struct.pack("<IiQQQQ", 37450, 2, d["beg"], d["end"], d["n"], 0)      # htslib's metadata pseudo-bin
        lin = d["lin"]
        for w in range(len(lin) - 1, -1, -1):
            if lin[w] is None:
                lin[w] = lin[w + 1] if w + 1 < len(lin) else 0
        out += struct.pack("<i", len(lin)) + struct.pack("<%dQ" % len(lin), *lin)
    with open(path + ".gz.tbi", "wb") as wf:
        for i in range(0, len(out), blk):
            wf.write(bgzf_compress_block(bytes(out[i:i + blk])))
        wf.write(eof)
    os.remove(path)


def build_freqb_parser():
    """Flags and defaults of `ccsmeth call_freqb` (ccsmeth.py:460-556)."""
    import argparse
    p = argparse.ArgumentParser(prog="ccsmeth_amd call_freqb", description="call modification frequencies from an aligned modbam")
    p.add_argument('--threads', type=int, default=5)
    p.add_argument('--input_bam', type=str, required=True,
                   help='aligned modbam with MM/ML tags, coordinate-sorted (every contig\'s records together: the file is streamed once, contig by '
                        'contig; an unsorted file is rejected)')
    p.add_argument('--ref', type=str, required=True)
    p.add_argument('--contigs', type=str, default=None)
    p.add_argument('--chunk_len', type=int, default=500000)
    p.add_argument('--output', '-o', type=str, required=True)
    p.add_argument('--bed', action='store_true', default=False)
    p.add_argument('--sort', action='store_true', default=False)
    p.add_argument('--gzip', action='store_true', default=False)
    p.add_argument('--modtype', type=str, default="5mC", choices=["5mC"])
    p.add_argument('--call_mode', type=str, default="count", choices=["count", "aggregate"])
    p.add_argument('--prob_cf', type=float, default=0.0)
    p.add_argument('--no_amb_cov', action="store_true", default=False)
    p.add_argument('--hap_tag', type=str, default="HP")
    p.add_argument('--mapq', type=int, default=1)
    p.add_argument('--identity', type=float, default=0.0)
    p.add_argument('--no_supplementary', action="store_true", default=False)
    p.add_argument('--motifs', type=str, default='CG')
    p.add_argument('--mod_loc', type=int, default=0)
    p.add_argument('--no_comb', action="store_true", default=False)
    p.add_argument('--refsites_only', action='store_true', default=False)
    p.add_argument('--refsites_all', action='store_true', default=False)
    p.add_argument('--no_hap', action="store_true", default=False)
    p.add_argument('--base_clip', type=int, default=0)
    p.add_argument('--aggre_model', '-m', type=str, default=None)
    p.add_argument('--model_type', type=str, default="attbigru", choices=["attbilstm", "attbigru"])
    p.add_argument('--seq_len', type=int, default=11)
    p.add_argument('--class_num', type=int, default=1)
    p.add_argument('--layer_rnn', type=int, default=1)
    p.add_argument('--hid_rnn', type=int, default=32)
    p.add_argument('--bin_size', type=int, default=20)
    p.add_argument('--cov_cf', type=int, default=4)
    p.add_argument('--only_close', action="store_true", default=False)
    p.add_argument('--discrete', action="store_true", default=False)
    p.add_argument('--tseed', type=int, default=1234)
    p.add_argument('--device', type=int, default=0, help="(this build) GPU of the aggregate model")
    return p


def main(argv=None):
    args = build_freqb_parser().parse_args(argv)
    call_mods_frequency_from_bamfile(args)
