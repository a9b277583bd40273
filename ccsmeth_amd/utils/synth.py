"""Synthetic weights / 21-mer feature batches for tests and bench.  NumPy only.

The generators are this repo's own (numpy.random.default_rng); the reference has none.
"""
import numpy as np

SEQ_LEN = 21


def codecv1_lut():
    """PacBio CodecV1 frame decode table (same values as reference utils/process_utils.py:426-449)."""
    lut = np.empty(256, dtype=np.int64)
    i = np.arange(64)
    lut[0:64] = i
    lut[64:128] = 64 + 2 * i
    lut[128:192] = 192 + 4 * i
    lut[192:256] = 448 + 8 * i
    return lut


def state_dict_shapes(seq_len=21, num_layers=3, num_classes=2, hidden=256, n_embed=8, n_vocab=5, feas_ccs=3):
    """Key -> shape of ModelAttRNN(attbigru2s).state_dict() (reference models.py:18-66)."""
    shapes = {"embed.weight": (n_vocab, n_embed)}
    for layer in range(num_layers):
        k_in = n_embed + feas_ccs if layer == 0 else 2 * hidden
        for sfx in ("", "_reverse"):
            shapes[f"rnn.weight_ih_l{layer}{sfx}"] = (3 * hidden, k_in)
            shapes[f"rnn.weight_hh_l{layer}{sfx}"] = (3 * hidden, hidden)
            shapes[f"rnn.bias_ih_l{layer}{sfx}"] = (3 * hidden,)
            shapes[f"rnn.bias_hh_l{layer}{sfx}"] = (3 * hidden,)
    shapes["_att3.Wa.weight"] = (hidden, 2 * hidden)
    shapes["_att3.Ua.weight"] = (hidden, 2 * hidden)
    shapes["_att3.va.weight"] = (1, hidden)
    shapes["fc1.weight"] = (num_classes, 4 * hidden)
    shapes["fc1.bias"] = (num_classes,)
    return shapes


def feas_ccs_of(is_npass=True, is_stds=False, is_sn=False, is_map=False):
    """Columns after the 8 embedding dims of a row of the layer-0 input (reference models.py:39-47)."""
    return 2 + int(bool(is_npass)) + 2 * int(bool(is_stds)) + 4 * int(bool(is_sn)) + int(bool(is_map))


def synth_extras(n, seed, is_stds=False, is_sn=False, is_map=False, seq_len=21):
    """The optional feature planes of the is_stds / is_sn / is_map model variants, one dict per strand: ipd_std, pw_std (N, 21) >= 0,
    sn (N, 4) signal-to-noise ratios, map (N, 21) in {0, 1}."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(2):
        d = {}
        if is_stds:
            d["ipd_std"] = rng.gamma(2.0, 0.4, size=(n, seq_len)).astype(np.float32)
            d["pw_std"] = rng.gamma(2.0, 0.4, size=(n, seq_len)).astype(np.float32)
        if is_sn:
            d["sn"] = np.round(rng.uniform(4.0, 18.0, size=(n, 4)), 6).astype(np.float32)
        if is_map:
            d["map"] = (rng.random(size=(n, seq_len)) < 0.9).astype(np.float32)
        out.append(d)
    return tuple(out)


def synth_weights(seed, num_layers=3, hidden=256, fc_bias=True, feas_ccs=3):
    """Random-init weights with the reference's init ranges: GRU / Linear default U(-1/sqrt(fan), 1/sqrt(fan)),
    embed and fc1.weight U(-0.1, 0.1) (models.py:71-75).  fc_bias=True draws a small non-zero fc1.bias so
    that parity tests exercise it (the reference zero-inits it; trained checkpoints do not keep it zero)."""
    rng = np.random.default_rng(seed)
    out = {}
    for key, shape in state_dict_shapes(num_layers=num_layers, hidden=hidden, feas_ccs=feas_ccs).items():
        if key.startswith("rnn."):
            bound = 1.0 / np.sqrt(hidden)
        elif key.startswith("_att3."):
            bound = 1.0 / np.sqrt(shape[-1])
        elif key == "fc1.bias":
            bound = 0.05 if fc_bias else 0.0
        else:
            bound = 0.1
        out[key] = rng.uniform(-bound, bound, size=shape).astype(np.float32)
    return out


def synth_weights_heavy(seed, outliers=8, outlier_scale=50.0, bias_shift=6.0):
    """A deliberately hostile checkpoint for the split-operand arithmetic: GRU matrices with Student-t (3 dof) entries at the
    synthetic scale, a few entries per matrix blown up by `outlier_scale`, and biases shifted by +-`bias_shift` on half of the units
    (gates driven into saturation).  Everything else as synth_weights."""
    w = synth_weights(seed)
    rng = np.random.default_rng(seed + 1000003)
    out = {}
    for k, v in w.items():
        if k.startswith("rnn.weight"):
            t = rng.standard_t(3, size=v.shape).astype(np.float32) * np.abs(v).mean()
            idx = rng.integers(0, t.size, size=outliers)
            t.reshape(-1)[idx] *= outlier_scale
            out[k] = t
        elif k.startswith("rnn.bias"):
            out[k] = (v + rng.choice([-bias_shift, 0.0, 0.0, bias_shift], size=v.shape)).astype(np.float32)
        else:
            out[k] = v
    return out


def synth_sites(n, seed, pseudo_read=15000):
    """n CpG sites of synthetic 21-mer features, both strands (SURVEY.md §8d recipe).

    Returns dict: kmer1/kmer2 uint8 (n,21) codes 0..4 with index 10 = C(1), 11 = G(2) and 0.1 % N(4) elsewhere;
    ipd1/pw1/ipd2/pw2 float32 (n,21) = CodecV1-decoded clipped-gamma codes, z-scored per 15 kb pseudo-read,
    rounded to 6 dp; npass1/npass2 float32 (n,) integer-valued U[3,30]."""
    rng = np.random.default_rng(seed)
    lut = codecv1_lut()
    d = {}
    for s in (1, 2):
        kmer = rng.integers(0, 4, size=(n, SEQ_LEN), dtype=np.uint8)
        nmask = rng.random((n, SEQ_LEN)) < 0.001
        kmer[nmask] = 4
        kmer[:, 10] = 1
        kmer[:, 11] = 2
        d[f"kmer{s}"] = kmer
        for name in ("ipd", "pw"):
            total = n * SEQ_LEN
            codes = np.clip(rng.gamma(2.0, 12.0, size=total), 0, 255).astype(np.int64)
            frames = lut[codes].astype(np.float64)
            out = np.empty(total, dtype=np.float64)
            for a in range(0, total, pseudo_read):
                seg = frames[a:a + pseudo_read]
                sd = seg.std()
                out[a:a + pseudo_read] = 0.0 if sd == 0 else (seg - seg.mean()) / sd
            d[f"{name}{s}"] = np.around(out, 6).astype(np.float32).reshape(n, SEQ_LEN)
        d[f"npass{s}"] = rng.integers(3, 31, size=n).astype(np.float32)
    return d


def synth_h0(n, seed, num_layers=3, hidden=256):
    """Explicit N(0,1) initial states for both strands: two arrays (2*num_layers, n, hidden) float32
    (reference layout models.py:77-87; strand-1 draw first)."""
    rng = np.random.default_rng(seed)
    h1 = rng.standard_normal((2 * num_layers, n, hidden), dtype=np.float32)
    h2 = rng.standard_normal((2 * num_layers, n, hidden), dtype=np.float32)
    return h1, h2


def synth_pileup(n_sites, seed, cov_lo=2, cov_hi=60):
    """Synthetic CpG pile-up for the aggregate model: reference positions with gaps ~ U[2,400],
    per-site coverage ~ U[cov_lo, cov_hi], per-read methylation probabilities ~ Beta(0.3, 0.3) quantised to ML bytes
    (floor(p * 256), what a modbam carries).  Returns dict(pos int64 (n,), ml list of uint8 arrays)."""
    rng = np.random.default_rng(seed)
    pos = np.cumsum(rng.integers(2, 401, size=n_sites)).astype(np.int64) + 10000
    cov = rng.integers(cov_lo, cov_hi + 1, size=n_sites)
    ml = [np.minimum(np.floor(rng.beta(0.3, 0.3, size=c) * 256), 255).astype(np.uint8) for c in cov]
    return dict(pos=pos, ml=ml)


def sample_index(name, size, k=1024):
    """Seeded sample of k flat indices of a tensor called `name` (all of them when size <= k): which entries of the big
    gradient / parameter tensors the training fixtures store (tests/golden/make_train_golden.py)."""
    if size <= k:
        return np.arange(size)
    h = 0
    for ch in name:
        h = (h * 131 + ord(ch)) % 1000000007
    return np.sort(np.random.default_rng(h).choice(size, k, replace=False))


def synth_aligned_reads(seed, n=14):
    """Aligned HiFi records for the --mode align fixtures: [(name, flag, mapq, cigar [(op, len)], SEQ as stored, fi, ri, fp, rp,
    fn, rn)] with soft / hard clips, = / X operators, all flag classes and a spread of MAPQ."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        body = int(rng.integers(150, 500))
        lead, trail = int(rng.choice([0, 0, 15, 40])), int(rng.choice([0, 0, 11, 33]))
        cigar = []
        if i % 5 == 1:
            cigar.append((5, 9))
        if lead:
            cigar.append((4, lead))
        a = body // 3
        cigar += [(0 if i % 3 else 7, a), (1, 2), (0, a), (2, 4), (8 if i % 4 == 0 else 0, body - 2 * a - 2)]
        if trail:
            cigar.append((4, trail))
        if i % 5 == 1:
            cigar.append((5, 4))
        L = lead + body + trail
        seq = rng.choice(list("ACGT"), size=L)
        for j in range(3, L - 1, 13):
            seq[j], seq[j + 1] = "C", "G"
        flag = [0, 16, 0, 16, 0x800, 0x800 | 16, 0x100, 0x400, 0x4, 0, 16, 0, 16, 0][i % 14]
        mapq = int([60, 60, 0, 5, 60, 60, 60, 60, 0, 1, 20, 60, 60, 3][i % 14])
        kin = lambda: rng.integers(0, 256, L).astype(np.uint8)  # noqa: E731
        out.append(("a%d" % i, flag, mapq, cigar, "".join(seq), kin(), kin(), kin(), kin(), int(rng.integers(3, 30)), int(rng.integers(3, 30))))
    return out


def synth_labeled_sites(n, seed, noise=0.04):
    """n sites with a PLANTED methylation signal, for training checkpoints whose weights have real structure (not the one-feature toy
    label of the early tests): label 1 sites carry an IPD / PW shift at window positions 8..12 of both strands whose size depends on the
    site (log-normal amplitude), on the strand and on the base in front of the C; `noise` of the labels are flipped.
    -> (sites dict as synth_sites, labels int64 (n,))."""
    d = synth_sites(n, seed)
    rng = np.random.default_rng(seed + 7919)
    y = (rng.random(n) < 0.5).astype(np.int64)
    amp = np.exp(rng.normal(0.0, 0.45, n)) * y
    pat_i = np.array([0.25, 0.55, 1.30, 0.80, 0.30])
    pat_p = np.array([0.10, -0.20, 0.45, 0.30, 0.00])
    for s, k in ((1, 1.0), (2, 0.7)):
        ctx = 1.0 + 0.35 * (d[f"kmer{s}"][:, 9].astype(np.float64) - 1.5) / 1.5            # the base in front of the C modulates the shift
        sh = (amp * ctx * k)[:, None]
        d[f"ipd{s}"][:, 8:13] = np.around(d[f"ipd{s}"][:, 8:13] + sh * pat_i + 0.15 * sh * rng.standard_normal((n, 5)), 6).astype(np.float32)
        d[f"pw{s}"][:, 8:13] = np.around(d[f"pw{s}"][:, 8:13] + sh * pat_p, 6).astype(np.float32)
    flip = rng.random(n) < noise
    return d, np.where(flip, 1 - y, y).astype(np.int64)
