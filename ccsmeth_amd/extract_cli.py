"""`python -m ccsmeth_amd extract` — HiFi BAM with kinetics -> the 22-column feature table `trainm` reads
(reference ccsmeth/extract_features.py:261-466, 538-610; flags of ccsmeth.py `extract`).

Host code (NumPy mirror of the per-read extraction, ccsmeth_amd/extract_features.py): this is the training-data side of the
path, not the inference hot path (call_mods extracts on the GPU).  Rows and their text are those of the reference:
chrom, pos, strand, holeid, loc, fkmer, fn, f_ipd x21, ".", f_pw x21, ".", ".", ".", rkmer, rn, r_ipd x21, ".", r_pw x21, ".", ".",
".", label — chrom / pos / strand are "." / -1 / "." in denovo mode and the reference position of the site's C in align mode."""
import argparse
import gzip
import os
import sys
import time

import numpy as np

from . import extract_features as ef
from .bamio import BamReader

DEFAULT_REF_LOC = -1                     # utils/process_utils.py default_ref_loc


def build_parser():
    p = argparse.ArgumentParser(prog="ccsmeth_amd extract", description="extract features from a HiFi BAM with kinetics")
    p.add_argument("--input", "-i", type=str, required=True)
    p.add_argument("--holeids_e", type=str, default=None)
    p.add_argument("--holeids_ne", type=str, default=None)
    p.add_argument("--output", "-o", type=str, required=False)
    p.add_argument("--gzip", action="store_true", default=False)
    p.add_argument("--mode", type=str, default="denovo", choices=["denovo", "align"])
    p.add_argument("--seq_len", type=int, default=21)
    p.add_argument("--motifs", type=str, default="CG")
    p.add_argument("--mod_loc", type=int, default=0)
    p.add_argument("--methy_label", type=int, choices=[1, 0], default=1)
    p.add_argument("--norm", type=str, choices=["zscore", "min-mean", "min-max", "mad", "none"], default="zscore")
    p.add_argument("--no_decode", action="store_true", default=False)
    p.add_argument("--path_to_samtools", type=str, default=None)
    p.add_argument("--holes_batch", type=int, default=50)
    p.add_argument("--is_sn", type=str, default="no")
    p.add_argument("--is_map", type=str, default="no")
    p.add_argument("--ref", type=str, required=False)
    p.add_argument("--mapq", type=int, default=1)
    p.add_argument("--identity", type=float, default=0.0)
    p.add_argument("--no_supplementary", action="store_true", default=False)
    p.add_argument("--skip_unmapped", type=str, default="yes")
    p.add_argument("--threads", type=int, default=5)
    return p


def _yes(v):
    return str(v).lower() in ("yes", "true", "t", "1")


def check_scope(args):
    if args.seq_len % 2 == 0:
        raise ValueError("--seq_len must be odd")                                    # extract_features.py:549-550
    if args.mode == "align" and args.ref is None:
        raise ValueError("--ref must be provided when using align mode!")            # :554-555
    if args.motifs.upper() != "CG" or args.mod_loc != 0:
        raise ValueError("this build implements --motifs CG --mod_loc 0")
    if args.norm not in ("zscore", "none"):
        raise ValueError("this build implements --norm zscore | none")
    if _yes(args.is_sn) or _yes(args.is_map):
        raise ValueError("this build implements --is_sn no --is_map no")


def get_q2tloc_from_cigar(cigar, strand, seq_len):
    """process_utils.py:190-226: reference offset of every aligned query base (-1 inserted, -2 never reached), in read
    direction (CIGAR reversed for the reverse strand); soft / hard clips are not walked."""
    q2r = np.full(seq_len + 1, -2, dtype=np.int32)
    r = q = 0
    for op, ln in (cigar if strand == 1 else cigar[::-1]):
        if op == 1:
            q2r[q:q + ln] = -1
            q += ln
        elif op in (2, 3):
            r += ln
        elif op in (0, 7, 8):
            q2r[q:q + ln] = np.arange(r, r + ln)
            q += ln
            r += ln
    q2r[q] = r
    if q2r[-1] == -2:
        raise ValueError("Invalid cigar string encountered. Reference length: {}  Cigar implied reference length: {}".format(seq_len, r))
    return q2r


def features_of_record(rec, ref_names, args, holeids_e=None, holeids_ne=None):
    """extract_features_from_double_strand_read (extract_features.py:261-406) for one bamio.BamRecord -> list of rows."""
    from .call_mods import _cigar_align_info
    name = rec.query_name
    if holeids_e is not None and name not in holeids_e:
        return []
    if holeids_ne is not None and name in holeids_ne:
        return []
    align = args.mode == "align"
    if align:
        if rec.flag & (0x4 | 0x100 | 0x400):
            return []
        if args.no_supplementary and rec.flag & 0x800:
            return []
        if rec.mapq < args.mapq:
            return []
    qs, qe, ident = _cigar_align_info(rec.cigar, len(rec.seq))
    if align and ident < args.identity:
        return []
    try:
        fi, ri, fp, rp = (np.asarray(rec.get_tag(t)) for t in ("fi", "ri", "fp", "rp"))
    except KeyError:
        return []
    try:
        fn, rn = rec.get_tag("fn"), rec.get_tag("rn")
    except KeyError:
        fn = rn = 0
    seq = rec.get_forward_sequence()
    arr = ef.extract_read_arrays(seq, fi, ri, fp, rp, args.seq_len, args.no_decode, args.norm)
    if arr is None:
        return []
    n = len(seq)
    reverse = rec.is_reverse
    seq_start, seq_end = (n - qe, n - qs) if reverse else (qs, qe)
    q2r = None
    if align:
        q2r = get_q2tloc_from_cigar(list(rec.cigar), -1 if reverse else 1, seq_end - seq_start)
        ref_start = rec.pos
        ref_end = rec.pos + sum(ln for op, ln in rec.cigar if op in (0, 2, 3, 7, 8))
        chrom = ref_names[rec.ref_id]
    rows = []
    for i in range(len(arr["loc"])):
        loc = int(arr["loc"][i])
        c, pos, strand = ".", DEFAULT_REF_LOC, "."
        if q2r is not None:
            c, strand = chrom, "-" if reverse else "+"
            if seq_start <= loc < seq_end:
                off = int(q2r[loc - seq_start])
                if off != -1:
                    pos = ref_end - 1 - off if reverse else off + ref_start
            elif _yes(args.skip_unmapped):
                continue
        rows.append([c, pos, strand, name, loc, arr["fkmer_ascii"][i].tobytes().decode("ascii"), fn, arr["fipd"][i], ".", arr["fpw"][i], ".",
                     ".", ".", arr["rkmer_ascii"][i].tobytes().decode("ascii"), rn, arr["ripd"][i], ".", arr["rpw"][i], ".", ".", ".",
                     args.methy_label])
    return rows


def _features_to_str(row):
    """extract_features.py:434-466 (str() of every float64, as the reference prints them)."""
    f = lambda a: ",".join([str(x) for x in a]) if type(a) is not str else "."  # noqa: E731
    return "\t".join([row[0], str(row[1]), row[2], row[3], str(row[4]), row[5], str(row[6]), f(row[7]), f(row[8]), f(row[9]), f(row[10]),
                      f(row[11]), f(row[12]), row[13], str(row[14]), f(row[15]), f(row[16]), f(row[17]), f(row[18]), f(row[19]), f(row[20]),
                      str(row[21])])


def extract_hifireads_features(args, log=sys.stderr):
    from .call_mods import _get_holes
    t0 = time.time()
    if not os.path.exists(args.input):
        raise IOError("input file does not exist!")                                  # :543-544
    check_scope(args)
    out = os.path.abspath(args.output) if args.output else os.path.splitext(args.input)[0] + ".features.tsv"   # :51-57
    holeids_e = None if args.holeids_e is None else _get_holes(args.holeids_e)
    holeids_ne = None if args.holeids_ne is None else _get_holes(args.holeids_ne)
    if args.gzip and not out.endswith(".gz"):
        out += ".gz"
    wf = gzip.open(out, "wt") if args.gzip else open(out, "w")
    n_reads = n_sites = 0
    with BamReader(args.input) as rd, wf:
        names = [r[0] for r in rd.references]
        for rec in rd:
            rows = features_of_record(rec, names, args, holeids_e, holeids_ne)
            n_reads += 1
            n_sites += len(rows)
            for row in rows:
                wf.write(_features_to_str(row) + "\n")
    log.write("[main]extract_features_hifi costs %.1f seconds.. (%d reads, %d sites) -> %s\n" % (time.time() - t0, n_reads, n_sites, out))
    return dict(reads=n_reads, sites=n_sites, output=out)


def main(argv=None):
    extract_hifireads_features(build_parser().parse_args(argv))
