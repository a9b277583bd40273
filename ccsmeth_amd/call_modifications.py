"""Host mirror of the batching / per-site result functions of reference ccsmeth/call_modifications.py on the
call_mods hot path: `_batch_feature_list2s` (:73-123) and `_call_mods2s` (:170-227), plus an array-based fast form that
skips the Python per-site lists.  The model call goes to libccsm (HIP); there is no CPU path."""
import numpy as np

from .utils.process_utils import base2code_dna

COALESCE_SITES = 24576          # sites per launch when _call_mods2s coalesces the reference's batches: four full rounds of 256 workgroups x 96 strand rows


def _batch_feature_list2s(feature_list):
    """22-field rows -> the reference's 18-tuple of parallel lists (call_modifications.py:73-123)."""
    sampleinfo = []
    cols = [[] for _ in range(16)]
    labels = []
    for featureline in feature_list:
        chrom, abs_loc, strand, holeid, loc, \
            kmer_seq, kmer_pass, kmer_ipdm, kmer_ipds, kmer_pwm, kmer_pws, kmer_sn, kmer_map, \
            kmer_seq2, kmer_pass2, kmer_ipdm2, kmer_ipds2, kmer_pwm2, kmer_pws2, kmer_sn2, kmer_map2, \
            label = featureline
        sampleinfo.append("\t".join(map(str, [chrom, abs_loc, strand, holeid, loc])))
        for off, (seq, npass, ipdm, ipds, pwm, pws, sn, mp) in enumerate((
                (kmer_seq, kmer_pass, kmer_ipdm, kmer_ipds, kmer_pwm, kmer_pws, kmer_sn, kmer_map),
                (kmer_seq2, kmer_pass2, kmer_ipdm2, kmer_ipds2, kmer_pwm2, kmer_pws2, kmer_sn2, kmer_map2))):
            o = off * 8
            cols[o + 0].append(np.array([base2code_dna[x] for x in seq]))
            cols[o + 1].append(np.array([npass] * len(seq)))
            cols[o + 2].append(np.array(ipdm, dtype=float))
            cols[o + 3].append(np.array(ipds, dtype=float) if type(ipds) is not str else 0)
            cols[o + 4].append(np.array(pwm, dtype=float))
            cols[o + 5].append(np.array(pws, dtype=float) if type(pws) is not str else 0)
            cols[o + 6].append(np.array(sn, dtype=float) if type(sn) is not str else 0)
            cols[o + 7].append(np.array(mp, dtype=float) if type(mp) is not str else 0)
        labels.append(label)
    return (sampleinfo, *cols, labels)


def prob1_norm_round6(probs):
    """Per site, in float32: round(prob_1 / (prob_0 + prob_1), 6) with NumPy's float32 scalar rounding
    (call_modifications.py:217-224)."""
    probs = np.asarray(probs, dtype=np.float32)
    return np.round(probs[:, 1] / (probs[:, 0] + probs[:, 1]), 6).astype(np.float32)


def _call_mods2s(features_batch, model, batch_size, device=0, h0_provider=None):
    """call_modifications.py:170-227: chunk the sites into `batch_size`, run the model, return
    ([(holeid, loc, prob_1_norm float32)], batch_num).  `h0_provider(batch_index, n)` (absent in the reference) pins the
    initial states for parity runs; default = the model's device RNG."""
    sampleinfo, fkmers, fpasss, fipdms, fipdsds, fpwms, fpwsds, fsns, fmaps, \
        rkmers, rpasss, ripdms, ripdsds, rpwms, rpwsds, rsns, rmaps, _ = features_batch
    pred_info = []
    batch_num = 0
    # The reference cuts a hole-batch into model calls of `batch_size` sites (default 512: 11 of the 256 workgroups a launch of this
    # library can hold).  All sites of the hole-batch are known before the first call and, with device-drawn initial states, a site's
    # result does not depend on how the calls are cut (its random stream is keyed by its running index; every launch form computes the same
    # bits), so against this library's model the loop below runs in launches of >= COALESCE_SITES sites and only COUNTS the reference's
    # batches.  Pinned initial states (h0_provider: parity runs) keep the reference's cut.
    coalesce = h0_provider is None and getattr(model, "coalesces_calls", False)
    step = max(int(batch_size), COALESCE_SITES) if coalesce else batch_size
    n_ref_batches = -(-len(sampleinfo) // int(batch_size)) if len(sampleinfo) else 0
    for i in np.arange(0, len(sampleinfo), step):
        s, e = i, i + step
        b_sampleinfo = sampleinfo[s:e]
        if len(b_sampleinfo) == 0:
            continue
        f32 = lambda a: np.asarray(np.array(a[s:e]), dtype=np.float32)  # noqa: E731  (FloatTensor(np.array(...)))
        h0 = h0_provider(batch_num, len(b_sampleinfo)) if h0_provider is not None else None
        _, vlogits = model(f32(fkmers), f32(fpasss), f32(fipdms), f32(fipdsds), f32(fpwms), f32(fpwsds), f32(fsns), f32(fmaps),
                           f32(rkmers), f32(rpasss), f32(ripdms), f32(ripdsds), f32(rpwms), f32(rpwsds), f32(rsns), f32(rmaps),
                           h0=h0)
        logits = np.asarray(vlogits.cpu().numpy() if hasattr(vlogits, "cpu") else vlogits)
        p1 = prob1_norm_round6(logits)
        for idx in range(len(b_sampleinfo)):
            words = b_sampleinfo[idx].split("\t")
            pred_info.append((words[3], int(words[4]), p1[idx]))
        batch_num += 1
    return pred_info, (n_ref_batches if coalesce else batch_num)
