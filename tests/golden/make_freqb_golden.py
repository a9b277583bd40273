#!/usr/bin/env python
"""Golden fixtures for `call_freqb` (the pile-up feed of the aggregate model), produced by the
REFERENCE's own functions.

    python tests/golden/make_freqb_golden.py

Writes tests/golden/freqb/{ref.fa, aligned.modbam.bam} (synthetic: this script is their generator) and
tests/golden/freqb_golden.json.gz: for a list of option sets, the text the reference writes to its .all / .hp1 / .hp2 files.

What runs from the reference (ccsmeth/call_mods_freq_bam.py, imported read-only from /root/reference):
_get_reference_chunks, _readmods_to_bed_of_one_region (record filters, _get_moddict -> _get_moddict_in_tags, the walk over
aligned pairs, strand combining, motif filter), _call_modfreq_of_one_region (count and aggregate mode, with the reference's
AggrAttRNN and its shipped checkpoint) and _write_one_line.  pysam is not installed here, so the AlignmentFile / AlignedSegment
objects those functions receive are duck-typed below from this repo's pure-Python BAM records; the parts of pysam that the
stand-ins restate are fetch() (records overlapping a region, file order), get_aligned_pairs(matches_only), get_cigar_stats(),
get_tag(), get_forward_sequence() and the flag properties.  modified_bases is reported empty so that the reference takes its
own MM/ML parser (_get_moddict_in_tags), which is the code under /root/reference that defines the expected values."""
import argparse
import gzip
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _ref_import import import_reference, REF_ROOT  # noqa: E402

import_reference()
import torch  # noqa: E402
import ccsmeth.call_mods_freq_bam as fb  # noqa: E402
from ccsmeth.utils.ref_reader import DNAReference  # noqa: E402
from ccsmeth.utils.process_utils import get_motif_seqs  # noqa: E402

from ccsmeth_amd import bamio  # noqa: E402

torch.set_num_threads(4)
OUT = os.path.join(HERE, "freqb")
CKPT = os.path.join(REF_ROOT, "models", "model_ccsmeth_5mCpG_aggregate_attbigru_b11.v2p.ckpt")
COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


# ---------------------------------------------------------------- synthetic genome and aligned modbam
def make_genome(rng):
    contigs = []
    for name, n in (("ctgB", 6000), ("ctgA", 5200), ("ctgC", 1200)):
        s = rng.choice(list("ACGT"), size=n, p=[0.22, 0.28, 0.28, 0.22]).tolist()
        for i in rng.choice(n - 1, size=n // 25, replace=False):       # CpG islands-ish
            s[i], s[i + 1] = "C", "G"
        contigs.append([name, s])
    contigs[0][1][2499], contigs[0][1][2500] = "C", "G"                # a CG across the 2500-base chunk boundary
    contigs[1][1][4999], contigs[1][1][5000] = "C", "G"
    contigs[1][1][100] = "N"
    return [(n, "".join(s)) for n, s in contigs]


def make_read(rng, tid, ref, idx):
    """One aligned record: a window of the reference with substitutions, insertions, deletions, clips."""
    n = len(ref)
    length = int(rng.integers(500, 1400))
    start = int(rng.integers(0, max(1, n - length)))
    end = min(n, start + length)
    cigar, q = [], []
    r = start
    if rng.random() < 0.3:
        k = int(rng.integers(1, 30))
        cigar.append((4, k))
        q += rng.choice(list("ACGT"), size=k).tolist()
    if idx % 37 == 5:
        cigar.insert(0, (5, 12))                                        # hard clip
    while r < end:
        run = int(min(end - r, rng.integers(20, 200)))
        seg = list(ref[r:r + run])
        for j in rng.choice(run, size=max(0, int(run * 0.02)), replace=False):
            seg[j] = str(rng.choice([c for c in "ACGT" if c != seg[j]]))
        op = 0 if idx % 5 else (7 if rng.random() < 0.8 else 8)         # some reads use = / X
        cigar.append((op, run))
        q += seg
        r += run
        if r >= end:
            break
        ev = rng.random()
        if ev < 0.35:
            k = int(rng.integers(1, 6))
            cigar.append((1, k))
            q += rng.choice(list("ACGT"), size=k).tolist()
        elif ev < 0.7:
            k = int(min(end - r - 1, rng.integers(1, 8)))
            if k > 0:
                cigar.append((2, k))
                r += k
        elif ev < 0.75 and end - r > 60:
            cigar.append((3, 40))
            r += 40
    if rng.random() < 0.3:
        k = int(rng.integers(1, 30))
        cigar.append((4, k))
        q += rng.choice(list("ACGT"), size=k).tolist()
    seq = "".join(q)
    reverse = bool(rng.random() < 0.5)
    flag = 16 if reverse else 0
    u = rng.random()
    if u < 0.04:
        flag |= 0x100
    elif u < 0.08:
        flag |= 0x400
    elif u < 0.14:
        flag |= 0x800
    elif u < 0.16:
        flag |= 0x4
    mapq = int(rng.choice([0, 1, 20, 60], p=[0.06, 0.04, 0.2, 0.7]))
    # MM/ML over the forward (as sequenced) strand
    fwd = revcomp(seq) if reverse else seq
    cs = [i for i, c in enumerate(fwd) if c == "C"]
    called = [k for k, i in enumerate(cs) if (i + 1 < len(fwd) and fwd[i + 1] == "G" and rng.random() < 0.92) or rng.random() < 0.01]
    tags = []
    hp = idx % 7
    if hp == 0:
        tags.append(("HP", "i", 1))
    elif hp == 1:
        tags.append(("HP", "C", 2))
    elif hp == 2:
        tags.append(("HP", "Z", "1"))
    elif hp == 3:
        tags.append(("HP", "i", 3))
    elif hp == 4:
        tags.append(("HP", "Z", "x"))
    elif hp == 5:
        tags.append(("HP", "s", 2))
    if idx % 29 == 7:
        called = []                                                      # "C+m?;" without positions
    if idx % 23 != 3:                                                    # a few reads carry no MM/ML at all
        deltas = [called[0]] + [called[k] - called[k - 1] - 1 for k in range(1, len(called))] if called else []
        mls = rng.choice([0, 1, 5, 30, 100, 127, 128, 129, 200, 250, 255], size=len(called)).astype(np.uint8)
        if idx % 31 == 11 and len(deltas) > 3:
            deltas[-1] += len(cs)                                        # points past the last C -> IndexError -> no calls
        if idx % 41 == 13 and len(mls) > 2:
            mls = mls[:-1]                                               # MM / ML length mismatch -> no calls
        style = "C+m" + ("?" if idx % 3 else ("." if idx % 2 else ""))
        tags.append(("MM", "Z", style + "".join("," + str(d) for d in deltas) + ";"))
        tags.append(("ML", "BC", mls))
    return bamio.BamRecord("m0/%d/ccs" % idx, flag=flag, ref_id=tid, pos=start, mapq=mapq, cigar=cigar, seq=seq, tags=tags)


def write_inputs(rng):
    os.makedirs(OUT, exist_ok=True)
    genome = make_genome(rng)
    with open(os.path.join(OUT, "ref.fa"), "w") as wf:
        for k, (name, s) in enumerate(genome):
            wf.write(">%s some description\n" % name)
            body = s.lower() if k == 2 else s
            for i in range(0, len(body), 60):
                wf.write(body[i:i + 60] + "\n")
    recs = []
    idx = 0
    for tid, (name, s) in enumerate(genome):
        for _ in range({0: 75, 1: 65, 2: 14}[tid]):
            recs.append(make_read(rng, tid, s, idx))
            idx += 1
    recs.sort(key=lambda r: (r.ref_id, r.pos))
    header = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (n, len(s)) for n, s in genome)
    path = os.path.join(OUT, "aligned.modbam.bam")
    with bamio.BamWriter(path, header, [(n, len(s)) for n, s in genome]) as bw:
        for r in recs:
            bw.write(r)
    return genome, recs


# ---------------------------------------------------------------- duck-typed pysam objects
class FakeSegment:
    def __init__(self, rec):
        self._r = rec
        self.query_name = rec.query_name
        self.is_unmapped = bool(rec.flag & 0x4)
        self.is_secondary = bool(rec.flag & 0x100)
        self.is_duplicate = bool(rec.flag & 0x400)
        self.is_supplementary = bool(rec.flag & 0x800)
        self.is_reverse = bool(rec.flag & 0x10)
        self.mapping_quality = rec.mapq
        self.modified_bases = None
        self.reference_start = rec.pos
        self.reference_end = rec.pos + sum(ln for op, ln in rec.cigar if op in (0, 2, 3, 7, 8))

    def get_cigar_stats(self):
        base = [0] * 11
        blocks = [0] * 11
        for op, ln in self._r.cigar:
            base[op] += ln
            blocks[op] += 1
        return base, blocks

    def get_tag(self, tag):
        v = self._r.get_tag(tag)
        return v

    def get_forward_sequence(self):
        return self._r.get_forward_sequence()

    def get_aligned_pairs(self, matches_only=False):
        out = []
        q, r = 0, self._r.pos
        for op, ln in self._r.cigar:
            if op in (0, 7, 8):
                out += [(q + i, r + i) for i in range(ln)]
                q += ln
                r += ln
            elif op in (1, 4):
                if not matches_only:
                    out += [(q + i, None) for i in range(ln)]
                q += ln
            elif op in (2, 3):
                if not matches_only:
                    out += [(None, r + i) for i in range(ln)]
                r += ln
        return out


class FakeAlignmentFile:
    def __init__(self, names, recs):
        self._names = names
        self._segs = [(r.ref_id, FakeSegment(r)) for r in recs]

    def fetch(self, contig=None, start=None, stop=None):
        if contig not in self._names:
            raise ValueError("invalid contig")
        tid = self._names.index(contig)
        for t, s in self._segs:
            if t == tid and s.reference_start < stop and max(s.reference_end, s.reference_start + 1) > start:
                yield s


# ---------------------------------------------------------------- option sets
def ns(**kw):
    base = dict(contigs=None, chunk_len=2500, bed=False, modtype="5mC", call_mode="count", prob_cf=0.0, no_amb_cov=False,
                hap_tag="HP", mapq=1, identity=0.0, no_supplementary=False, motifs="CG", mod_loc=0, no_comb=False,
                refsites_only=False, refsites_all=False, no_hap=False, base_clip=0, aggre_model=CKPT, model_type="attbigru",
                seq_len=11, class_num=1, layer_rnn=1, hid_rnn=32, bin_size=20, cov_cf=4, only_close=False, discrete=False,
                tseed=1234)
    base.update(kw)
    return base


CASES = {
    "count_default": ns(),
    "count_bed": ns(bed=True),
    "count_no_comb": ns(no_comb=True),
    "count_refsites_only": ns(refsites_only=True),
    "count_refsites_all": ns(refsites_all=True),
    "count_refsites_all_no_comb_clip": ns(refsites_all=True, no_comb=True, base_clip=7),
    "count_prob_cf": ns(prob_cf=0.3),
    "count_prob_cf_no_amb": ns(prob_cf=0.3, no_amb_cov=True),
    "count_clip_nohap": ns(base_clip=25, no_hap=True),
    "count_filters": ns(mapq=20, identity=0.975, no_supplementary=True),
    "count_contigs_chunk": ns(contigs="ctgA,ctgC", chunk_len=1000),
    "count_hp_tag_other": ns(hap_tag="XX"),
    "count_chg_motif": ns(motifs="CHG", refsites_only=True, no_comb=True),
    "aggregate_default": ns(call_mode="aggregate"),
    "aggregate_bed_cov6": ns(call_mode="aggregate", bed=True, cov_cf=6),
    "aggregate_no_comb_discrete": ns(call_mode="aggregate", no_comb=True, discrete=True),
    "aggregate_nohap_refsites_only": ns(call_mode="aggregate", no_hap=True, refsites_only=True, chunk_len=4000),
    "aggregate_only_close": ns(call_mode="aggregate", only_close=True),
}


def run_reference(case, dnacontigs, bam):
    args = argparse.Namespace(**case)
    motifs = get_motif_seqs(args.motifs)
    motifs_filter = motifs if (args.refsites_only or args.refsites_all) else None
    chunks = fb._get_reference_chunks(dnacontigs, args.contigs, args.chunk_len, args.motifs)
    outs = [io.StringIO() for _ in range(3)]
    for region in chunks:
        beds = fb._readmods_to_bed_of_one_region(bam, region, dnacontigs, motifs_filter, args)
        for wf, bed in zip(outs, beds):
            for item in bed:
                fb._write_one_line(item, wf, args.bed)
    return {"chunks": [[c[0], int(c[1]), int(c[2])] for c in chunks], "all": outs[0].getvalue(), "hp1": outs[1].getvalue(),
            "hp2": outs[2].getvalue()}


def main():
    rng = np.random.default_rng(20260928)
    genome, recs = write_inputs(rng)
    dnacontigs = DNAReference(os.path.join(OUT, "ref.fa")).getcontigs()
    assert [dnacontigs[n] for n, _ in genome] == [s for _, s in genome]
    bam = FakeAlignmentFile([n for n, _ in genome], recs)
    golden = {}
    for name, case in CASES.items():
        res = run_reference(case, dnacontigs, bam)
        opts = {k: v for k, v in case.items() if k != "aggre_model"}
        golden[name] = {"args": opts, **res}
        print(name, {k: len(res[k].splitlines()) for k in ("all", "hp1", "hp2")})
    with open(os.path.join(HERE, "freqb_golden.json.gz"), "wb") as raw:
        with gzip.GzipFile(fileobj=raw, mode="wb", mtime=0) as wf:
            wf.write(json.dumps(golden, indent=0).encode("ascii"))
    print("bam bytes", os.path.getsize(os.path.join(OUT, "aligned.modbam.bam")))


if __name__ == "__main__":
    main()
