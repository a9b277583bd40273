"""Per-read 21-mer kinetics feature extraction — host mirror of reference ccsmeth/extract_features.py:181-199, 261-406
(denovo mode, defaults: --norm zscore, --motifs CG, --mod_loc 0, --seq_len 21, is_sn/is_map no).

Vectorised NumPy (LUT gather, whole-read z-score, CG scan and window gather by fancy indexing) instead of the reference's
per-base list comprehensions; results are bit-exact: float64 mean / std(ddof=0) over the whole read via the same NumPy
reductions, `np.around(., 6)`, all-zero when std == 0.

`extract_read_arrays` produces the site features directly in the layout libccsm's ccsm_batch wants;
`extract_features_from_double_strand_read` wraps them into the reference's 22-field rows."""
import numpy as np

from .utils.process_utils import CODE2FRAMES, complement_seq, seq_to_codes

_ASCII_C, _ASCII_G = ord("C"), ord("G")


def _normalize_signals(signals, normalize_method="zscore"):
    """extract_features.py:181-199, every method: none | zscore | min-max | min-mean | mad (statsmodels.robust.scale.mad = median
    absolute deviation from the median / Phi^-1(3/4)).  float64 like the reference; a zero scale gives all zeros."""
    signals = np.asarray(signals)
    if normalize_method == "none":
        return np.around(signals, decimals=6)
    if normalize_method == "zscore":
        sshift, sscale = np.mean(signals), np.std(signals)
    elif normalize_method == "min-max":
        sshift, sscale = np.min(signals), np.max(signals) - np.min(signals)
    elif normalize_method == "min-mean":
        sshift, sscale = np.min(signals), np.mean(signals)
    elif normalize_method == "mad":
        med = np.median(signals)
        sshift, sscale = med, float(np.median(np.abs(signals - med)) / 0.6744897501960817)
    else:
        raise ValueError("")                                  # (the reference's own message)
    if sscale == 0.0:
        return np.zeros(len(signals), dtype=np.float64)
    return np.around((signals - sshift) / sscale, decimals=6)


def motif_locs_cg(seq_bytes):
    """Positions i with seq[i:i+2] == 'CG' (process_utils.py:122-137 for the symmetric motif CG, mod_loc 0)."""
    return np.flatnonzero((seq_bytes[:-1] == _ASCII_C) & (seq_bytes[1:] == _ASCII_G)) if len(seq_bytes) > 1 else np.empty(0, np.int64)


def count_kept_sites(seq_bytes, seq_len=21):
    """Number of CG sites of one read that pass the window test of extract_features.py:343-350."""
    n, nb = len(seq_bytes), (seq_len - 1) // 2
    locs = motif_locs_cg(np.asarray(seq_bytes))
    rl = n - 1 - (locs + 1)
    return int(np.count_nonzero((locs >= nb) & (locs < n - nb) & (rl >= nb) & (rl < n - nb)))


def extract_read_arrays(seq, fi, ri, fp, rp, seq_len=21, no_decode=False, norm="zscore"):
    """One double-strand HiFi read -> per-site arrays (n_sites may be 0), or None when the kinetics arrays do not match
    the sequence length (extract_features.py:320-325 -> read skipped).

    Returns dict: loc int64 (n,), fkmer/rkmer uint8 codes (n,21), fipd/fpw/ripd/rpw float64 (n,21) (rounded to 6 dp,
    exactly the reference's values), plus fkmer_ascii/rkmer_ascii uint8 (n,21)."""
    n = len(seq)
    fi, ri, fp, rp = (np.asarray(a) for a in (fi, ri, fp, rp))
    if not (len(fi) == n and len(fp) == n and len(ri) == n and len(rp) == n):
        return None
    def dec(a):
        a = a.astype(np.int64)
        return a if no_decode else CODE2FRAMES[a]
    ipd_f, ipd_r = _normalize_signals(dec(fi), norm), _normalize_signals(dec(ri), norm)   # ri/rp are NOT flipped (:314-319)
    pw_f, pw_r = _normalize_signals(dec(fp), norm), _normalize_signals(dec(rp), norm)
    sb = np.frombuffer(seq.encode("ascii"), dtype=np.uint8)
    rcb = np.frombuffer(complement_seq(seq).encode("ascii"), dtype=np.uint8)
    nb = (seq_len - 1) // 2
    locs = motif_locs_cg(sb)
    rev_loc_in_rev = n - 1 - (locs + 1)                    # rev_offset_loc = 1 for CG / mod_loc 0 (:341-346)
    keep = (locs >= nb) & (locs < n - nb) & (rev_loc_in_rev >= nb) & (rev_loc_in_rev < n - nb)   # :347
    locs, rl = locs[keep], rev_loc_in_rev[keep]
    win = np.arange(-nb, nb + 1)
    fidx = locs[:, None] + win[None, :]
    ridx = rl[:, None] + win[None, :]
    fk, rk = sb[fidx], rcb[ridx]
    return dict(loc=locs.astype(np.int64), fkmer_ascii=fk, rkmer_ascii=rk, fkmer=seq_to_codes(fk), rkmer=seq_to_codes(rk),
                fipd=ipd_f[fidx], fpw=pw_f[fidx], ripd=ipd_r[ridx], rpw=pw_r[ridx])


def extract_features_from_double_strand_read(seq_name, seq, fi, ri, fp, rp, fn, rn, seq_len=21, methy_label=1,
                                             no_decode=False, norm="zscore"):
    """Reference-shaped output (extract_features.py:394-405, denovo): a list of 22-field rows
    [chrom '.', pos -1, strand '.', holeid, loc, fkmer, fn, f_ipd, '.', f_pw, '.', '.', '.', rkmer, rn, r_ipd, '.',
    r_pw, '.', '.', '.', label]."""
    arr = extract_read_arrays(seq, fi, ri, fp, rp, seq_len, no_decode, norm)
    if arr is None:
        return []
    rows = []
    for i in range(len(arr["loc"])):
        rows.append([".", -1, ".", seq_name, int(arr["loc"][i]),
                     arr["fkmer_ascii"][i].tobytes().decode("ascii"), fn, arr["fipd"][i], ".", arr["fpw"][i], ".", ".", ".",
                     arr["rkmer_ascii"][i].tobytes().decode("ascii"), rn, arr["ripd"][i], ".", arr["rpw"][i], ".", ".", ".",
                     methy_label])
    return rows
