"""CPU oracle (NumPy restatement) of the ccsmeth `call_mods` attbigru2s hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ccsmeth_amd/ may import this module; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker.

Parity status: PINNED by outputs of the reference itself.  The reference ships no tests and no golden
vectors; this restatement is pinned against tests/golden/*.npz, which were produced by
importing /root/reference/ccsmeth in the build container (tests/golden/make_golden.py, h0 injected by
wrapping torch.randn) and are checked in tests/test_oracle_golden.py.

Third-party arithmetic: the GRU cell is torch.nn.GRU (ATen `_VF.gru`; reference pins torch>=1.2,<=2.1.0 in
requirements.txt:4, environment.yml:10); its published equations are restated in `gru_direction` below and
anchored on the reference's call sites models.py:54-55 (ctor) and models.py:125-130 (calls).

Every function cites the reference file:line it follows (paths relative to /root/reference/).
"""
import math

import numpy as np

# ccsmeth/utils/process_utils.py:64-73
N_VOCAB = 5
NEMBED_BASE = 8
# ccsmeth/utils/process_utils.py:26-29 — every IUPAC code other than ACGT maps to 4
BASE2CODE_DNA = {'A': 0, 'C': 1, 'G': 2, 'T': 3, 'N': 4, 'W': 4, 'S': 4, 'M': 4, 'K': 4, 'R': 4,
                 'Y': 4, 'B': 4, 'V': 4, 'D': 4, 'H': 4, 'Z': 4}


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def gru_direction(x, h0, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of one nn.GRU layer (batch_first).  x (N,L,K), h0 (N,H) -> out (N,L,H), h_n (N,H).

    torch.nn.GRU equations (gate row order [r; z; n], reference call site models.py:125-130):
      gi = x_t W_ih^T + b_ih ; gh = h W_hh^T + b_hh
      r = sigmoid(gi_r + gh_r); z = sigmoid(gi_z + gh_z); n = tanh(gi_n + r * gh_n)
      h' = (h - n) * z + n          (ATen gru_cell form of (1-z)*n + z*h)
    """
    n_b, seq_len, _ = x.shape
    hid = h0.shape[1]
    h = h0.astype(x.dtype, copy=True)
    out = np.empty((n_b, seq_len, hid), dtype=x.dtype)
    gi_all = x @ w_ih.T + b_ih  # (N,L,3H)
    steps = range(seq_len - 1, -1, -1) if reverse else range(seq_len)
    for t in steps:
        gi = gi_all[:, t, :]
        gh = h @ w_hh.T + b_hh
        r = _sigmoid(gi[:, :hid] + gh[:, :hid])
        z = _sigmoid(gi[:, hid:2 * hid] + gh[:, hid:2 * hid])
        n = np.tanh(gi[:, 2 * hid:] + r * gh[:, 2 * hid:])
        h = (h - n) * z + n
        out[:, t, :] = h
    return out, h


def bigru(x, h0, weights, num_layers, prefix="rnn."):
    """Stacked bidirectional GRU.  h0 (2*num_layers, N, H), index 2l = layer-l forward, 2l+1 = backward
    (models.py:77-87 init_hidden layout; torch h_0 convention).  Returns (out (N,L,2H), h_n (2*layers,N,H))."""
    h_n = []
    inp = x
    for layer in range(num_layers):
        outs = []
        for d, sfx in enumerate(("", "_reverse")):
            o, hn = gru_direction(inp, h0[2 * layer + d],
                                  weights[f"{prefix}weight_ih_l{layer}{sfx}"], weights[f"{prefix}weight_hh_l{layer}{sfx}"],
                                  weights[f"{prefix}bias_ih_l{layer}{sfx}"], weights[f"{prefix}bias_hh_l{layer}{sfx}"],
                                  reverse=bool(d))
            outs.append(o)
            h_n.append(hn)
        inp = np.concatenate(outs, axis=2)
    return inp, np.stack(h_n, 0)


def attention(last_hidden, enc_out, wa, ua, va):
    """Bahdanau attention pool, utils/attention.py:48-70.
    last_hidden (N,2H) query, enc_out (N,L,2H).  e = va . tanh(Wa q + Ua k); a = softmax_L(e); c = sum_t a_t k_t."""
    q = last_hidden @ wa.T                       # (N,A)            attention.py:69  Wa(last_hidden)
    k = enc_out @ ua.T                           # (N,L,A)          attention.py:69  Ua(encoder_outputs)
    e = np.tanh(q[:, None, :] + k) @ va.reshape(-1)  # (N,L)        attention.py:70
    e = e - e.max(axis=1, keepdims=True)
    a = np.exp(e)
    a = a / a.sum(axis=1, keepdims=True)         # attention.py:55 softmax over L
    ctx = np.einsum("nl,nlc->nc", a, enc_out)    # attention.py:57-58
    return ctx, a


def strand_input(embed_w, kmer, ipd, pw, npass, extra=None, features=(True, False, False, False)):
    """models.py:91-123: x = cat(embed[kmer.int()], ipd, pw [, npass] [, ipd_std, pw_std] [, sn expanded over L] [, map]) -> (N, L, C);
    features = (is_npass, is_stds, is_sn, is_map); the defaults (call_modifications.py:652-663) give C = 11."""
    is_npass, is_stds, is_sn, is_map = features
    kmer = np.asarray(kmer)
    idx = kmer.astype(np.int32)                  # models.py:91  kmer.int()  (truncation toward zero)
    emb = embed_w[idx]
    n_b, seq_len = idx.shape
    dt = embed_w.dtype
    feats = [emb, np.asarray(ipd, dt)[..., None], np.asarray(pw, dt)[..., None]]
    if is_npass:                                 # models.py:100-104
        npass = np.asarray(npass, dtype=dt)
        if npass.ndim == 1:
            npass = np.repeat(npass[:, None], seq_len, axis=1)   # call_modifications.py:96  [npass]*len(kmer)
        feats.append(npass[..., None])
    if is_stds:                                  # models.py:105-111
        feats += [np.asarray(extra["ipd_std"], dt)[..., None], np.asarray(extra["pw_std"], dt)[..., None]]
    if is_sn:                                    # models.py:112-116  sns.unsqueeze(1).expand(-1, L, -1)
        feats.append(np.repeat(np.asarray(extra["sn"], dt)[:, None, :], seq_len, axis=1))
    if is_map:                                   # models.py:117-121
        feats.append(np.asarray(extra["map"], dt)[..., None])
    return np.concatenate(feats, axis=2)


def attbigru2s_forward(weights, kmer1, ipd1, pw1, npass1, kmer2, ipd2, pw2, npass2, h0_1, h0_2,
                       num_layers=3, dtype=np.float64, extra=None, features=(True, False, False, False)):
    """ModelAttRNN(model_type="attbigru2s").forward, models.py:89-150, with h0 pinned (the reference draws
    torch.randn per strand, models.py:77-87,125-130 — strand 1 first).  Returns (logits (N,2), probs (N,2)).
    extra = (strand-1 dict, strand-2 dict) of the planes `features` = (is_npass, is_stds, is_sn, is_map) asks for."""
    w = {k: np.asarray(v, dtype=dtype) for k, v in weights.items()}
    ctxs = []
    extra = extra or (None, None)
    for kmer, ipd, pw, npass, h0, ex in ((kmer1, ipd1, pw1, npass1, h0_1, extra[0]), (kmer2, ipd2, pw2, npass2, h0_2, extra[1])):
        x = strand_input(w["embed.weight"], kmer, np.asarray(ipd, dtype), np.asarray(pw, dtype),
                         None if npass is None else np.asarray(npass, dtype), ex, features)
        out, h_n = bigru(x, np.asarray(h0, dtype), w, num_layers)
        # models.py:135-137: last layer's (fwd, bwd) final states -> (N, 2H)
        q = np.concatenate([h_n[2 * (num_layers - 1)], h_n[2 * (num_layers - 1) + 1]], axis=1)
        ctx, _ = attention(q, out, w["_att3.Wa.weight"], w["_att3.Ua.weight"], w["_att3.va.weight"])
        ctxs.append(ctx)
    feat = np.concatenate(ctxs, axis=1)                         # models.py:145
    logits = feat @ w["fc1.weight"].T + w["fc1.bias"]           # models.py:148 (dropout = identity in eval)
    m = logits.max(axis=1, keepdims=True)
    p = np.exp(logits - m)
    probs = p / p.sum(axis=1, keepdims=True)                    # models.py:150 Softmax(1)
    return logits, probs


def prob1_norm_round6(probs_f32):
    """call_modifications.py:217-224: per site, in float32, round(prob_1 / (prob_0 + prob_1), 6) using the NumPy
    float32 scalar __round__ (not float64)."""
    probs_f32 = np.asarray(probs_f32, dtype=np.float32)
    out = np.empty(probs_f32.shape[0], dtype=np.float32)
    for i in range(probs_f32.shape[0]):
        p0, p1 = probs_f32[i, 0], probs_f32[i, 1]
        out[i] = round(p1 / (p0 + p1), 6)
    return out


# ---------------------------------------------------------------------------------------------------------
# Host-side rows: feature extraction, batching, MM/ML encode.
# ---------------------------------------------------------------------------------------------------------

def codecv1_to_frame2():
    """utils/process_utils.py:426-449: 256-entry CodecV1 LUT.  0-63 -> id; 64-127 -> 64+2k; 128-191 -> 192+4k;
    192-255 -> 448+8k (max 952)."""
    lut = [0] * 256
    for i in range(64):
        lut[i] = i
        lut[64 + i] = 64 + 2 * i
        lut[128 + i] = 192 + 4 * i
        lut[192 + i] = 448 + 8 * i
    return lut


def complement_seq(seq):
    """utils/process_utils.py:106-118 (DNA): reverse complement with the IUPAC pair table of
    process_utils.py:12-15; any base outside the table -> 'N' (process_utils.py:100-103)."""
    iupac = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A', 'N': 'N', 'W': 'W', 'S': 'S', 'M': 'K', 'K': 'M',
             'R': 'Y', 'Y': 'R', 'B': 'V', 'V': 'B', 'D': 'H', 'H': 'D', 'Z': 'Z'}
    return ''.join(iupac.get(b, 'N') for b in reversed(seq))


def normalize_signals_zscore(signals):
    """extract_features.py:181-199 (normalize_method='zscore'): float64 mean/std(ddof=0) over the WHOLE read,
    all-zero if std == 0, np.around(., 6)."""
    signals = np.asarray(signals)
    sshift, sscale = np.mean(signals), np.std(signals)
    if sscale == 0.0:
        norm = [0.] * len(signals)
    else:
        norm = (signals - sshift) / sscale
    return np.around(norm, decimals=6)


def motif_sites(seq, motif="CG", mod_loc=0):
    """utils/process_utils.py:122-137 get_refloc_of_methysite_in_motif for one motif."""
    m = len(motif)
    return [i + mod_loc for i in range(0, len(seq) - m + 1) if seq[i:i + m] == motif]


def extract_read_features(seq, fi, ri, fp, rp, fn, rn, seq_len=21, motif="CG", mod_loc=0):
    """extract_features.py:261-406 in denovo mode with defaults (norm zscore, is_sn/is_map no, no_decode False).
    seq = forward sequence; fi/ri/fp/rp = uint8 CodecV1 codes (len == len(seq)); fn/rn = pass counts.
    Returns a list of per-site tuples (loc, fkmer, fn, f_ipd(21), f_pw(21), rkmer, rn, r_ipd(21), r_pw(21))."""
    lut = np.asarray(codecv1_to_frame2())
    n = len(seq)
    if not (len(fi) == n and len(fp) == n and len(ri) == n and len(rp) == n):
        return []                                              # extract_features.py:320-325
    ipd_f = normalize_signals_zscore(lut[np.asarray(fi, dtype=int)])   # :327-334
    ipd_r = normalize_signals_zscore(lut[np.asarray(ri, dtype=int)])   # ri/rp are NOT flipped (:314-319)
    pw_f = normalize_signals_zscore(lut[np.asarray(fp, dtype=int)])
    pw_r = normalize_signals_zscore(lut[np.asarray(rp, dtype=int)])
    seq_rc = complement_seq(seq)
    nb = (seq_len - 1) // 2
    rev_offset = (len(motif) - 1 - mod_loc) - mod_loc          # :341
    rows = []
    for loc in motif_sites(seq, motif, mod_loc):
        rev_loc = loc + rev_offset
        rl = n - 1 - rev_loc                                   # rev_loc_in_rev :346
        if nb <= loc < n - nb and nb <= rl < n - nb:           # :347
            rows.append((loc, seq[loc - nb:loc + nb + 1], fn, ipd_f[loc - nb:loc + nb + 1], pw_f[loc - nb:loc + nb + 1],
                         seq_rc[rl - nb:rl + nb + 1], rn, ipd_r[rl - nb:rl + nb + 1], pw_r[rl - nb:rl + nb + 1]))
    return rows


def convert_locs_to_mmtag(locs, seq_fwd, base="C"):
    """_bam2modbam.py:187-203: for each called C (sorted locs) its ordinal among ALL `base` in seq_fwd,
    delta-coded (first, o_i - 1 - o_{i-1}).  AssertionError if empty or the last loc is not a `base`."""
    assert len(locs) > 0
    all_locs = [i for i, b in enumerate(seq_fwd) if b == base]
    orders = [-1] * len(locs)
    oi = 0
    for bi in range(len(all_locs)):
        if oi >= len(locs):
            break
        if all_locs[bi] == locs[oi]:
            orders[oi] = bi
            oi += 1
    assert orders[-1] != -1
    mm = [orders[0]]
    for i in range(1, len(orders)):
        mm.append(orders[i] - 1 - orders[i - 1])
    return mm


def convert_probs_to_mltag(probs):
    """_bam2modbam.py:206-208: floor(p*256), 255 if p >= 1."""
    return [math.floor(p * 256) if p < 1 else 255 for p in probs]


# ---------------------------------------------------------------------------------------------------------
# Aggregate model (BASELINE config 5): per-site methylation frequency from pile-up histograms.
# ---------------------------------------------------------------------------------------------------------

def cal_mod_prob(ml_value):
    """call_mods_freq_bam.py:102-107: ML byte -> probability, round(ml/256 + 1e-6, 6), 0 for ml == 0."""
    return round(ml_value / float(256) + 0.000001, 6) if ml_value > 0 else 0


def normalized_histo(probs, cov_cf=4, binsize=20):
    """call_mods_freq_bam.py:221-237: 20-bin histogram over [0,1], divided by its L2 norm, rounded to 6 dp."""
    assert len(probs) >= cov_cf
    hist = np.histogram(probs, bins=binsize, range=[0, 1])[0]
    return np.round(hist / np.linalg.norm(hist), 6)


def aggregate_windows(refposes, histos, seq_len=11, only_close=False):
    """call_mods_freq_bam.py:270-290: zero-padded histogram windows (M,L,20) and the position feature (M,L): |pos - centre| with pad
    positions first-1000 / last+1000, or with only_close (:285-290) 1 where a window site lies exactly 2 bases after its predecessor."""
    refposes = np.asarray(refposes, dtype=np.int64)
    histos = np.asarray(histos, dtype=np.float64)
    m, pad = len(refposes), seq_len // 2
    hp = np.zeros((m + 2 * pad, histos.shape[1]), dtype=np.float64)
    hp[pad:pad + m] = histos
    pp = np.concatenate([np.full(pad, refposes[0] - 1000), refposes, np.full(pad, refposes[-1] + 1000)])
    idx = np.arange(m)[:, None] + np.arange(seq_len)[None, :]
    if only_close:
        pq = np.concatenate([np.full(pad + 1, refposes[0] - 1000), refposes, np.full(pad, refposes[-1] + 1000)])
        return hp[idx], (np.diff(pq) == 2).astype(np.int64)[idx]
    return hp[idx], np.abs(pp[idx] - refposes[:, None])


def aggr_attbigru_forward(weights, offsets, histos, h0, dtype=np.float64):
    """AggrAttRNN.forward, models.py:673-694: x = cat(histos, offsets) (histogram first, offset last), 1-layer BiGRU(H=32),
    attention with query = final states, fc1 (64 -> 1), no softmax.  h0 (2, M, 32)."""
    w = {k: np.asarray(v, dtype=dtype) for k, v in weights.items()}
    x = np.concatenate([np.asarray(histos, dtype), np.asarray(offsets, dtype)[..., None]], axis=2)
    out, h_n = bigru(x, np.asarray(h0, dtype), w, 1)
    q = np.concatenate([h_n[0], h_n[1]], axis=1)
    ctx, _ = attention(q, out, w["_att3.Wa.weight"], w["_att3.Ua.weight"], w["_att3.va.weight"])
    return ctx @ w["fc1.weight"].T + w["fc1.bias"]


def cal_modfreq_in_aggregate_mode(refposes, histos, weights, h0_normals, stream_pos=0, seq_len=11, batch_size=1024, only_close=False):
    """call_mods_freq_bam.py:265-305: batches of 1024, h0 = the next 64*B values of the seeded randn stream reshaped
    (2,B,32), output round(clip(y,0,1),6) as float32.  Returns (probs float32 (M,), new stream position)."""
    histos_mat, pos_mat = aggregate_windows(refposes, histos, seq_len, only_close)
    probs = []
    for s in range(0, len(histos_mat), batch_size):
        b_h = histos_mat[s:s + batch_size].astype(np.float32)
        b_p = pos_mat[s:s + batch_size].astype(np.float32)
        n = len(b_h)
        h0 = np.asarray(h0_normals[stream_pos:stream_pos + 64 * n], np.float32).reshape(2, n, 32)
        stream_pos += 64 * n
        y = aggr_attbigru_forward(weights, b_p, b_h, h0, dtype=np.float64).astype(np.float32)
        probs.append(np.round(np.clip(y, 0, 1), 6)[:, 0])
    return np.concatenate(probs).astype(np.float32), stream_pos
