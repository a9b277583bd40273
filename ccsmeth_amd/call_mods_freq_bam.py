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

    def forward_raw(self, refposes, histos):
        """fc1 outputs (M,) float32 for the sites of one call; advances the region's random stream by 64 values per site."""
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
    if only_close:
        raise ValueError("only_close is outside this build")
    y = model.forward_raw(refposes, np.stack(refposes_histos))
    return list(np.round(np.clip(y, 0, 1), 6))
