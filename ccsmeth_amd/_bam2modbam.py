"""MM / ML tag encoding of per-read CpG calls — host mirror of reference ccsmeth/_bam2modbam.py:187-226.

Same function names, argument meaning and error behaviour (AssertionError on an empty call list or a last location
that is not a C), integer-exact results.  NumPy implementation (searchsorted over the C positions) instead of the
reference's regex + Python scan."""
import math

import numpy as np

base = "C"  # the reference's module-level `base` (_bam2modbam.py:24)


def _convert_locs_to_mmtag(locs, seq_fwseq):
    """locs: sorted 0-based positions of called C's in the forward read sequence.  Returns the MM delta list:
    ordinal of the first called C among all C's, then (ordinal_i - 1 - ordinal_{i-1}).  (_bam2modbam.py:187-203)

    The reference walks all C's once with a moving pointer, so a location that is not a C (or is out of order)
    stalls the pointer and leaves the LAST order at -1 -> AssertionError; reproduced here."""
    assert len(locs) > 0
    seq_bytes = np.frombuffer(seq_fwseq.encode("ascii"), dtype=np.uint8)
    all_c = np.flatnonzero(seq_bytes == ord(base))
    locs_a = np.asarray(locs, dtype=np.int64)
    orders = np.full(len(locs_a), -1, dtype=np.int64)
    # emulate the single forward scan: each loc can only match a C at or after the previous match
    idx = np.searchsorted(all_c, locs_a)
    prev = -1
    for i in range(len(locs_a)):
        j = int(idx[i])
        if j < len(all_c) and all_c[j] == locs_a[i] and j > prev:
            orders[i] = j
            prev = j
        else:
            break          # the reference's pointer never advances past an unmatched location
    assert orders[-1] != -1
    mm = [int(orders[0])]
    mm.extend(int(orders[i] - 1 - orders[i - 1]) for i in range(1, len(orders)))
    return mm


def _convert_probs_to_mltag(probs):
    """ML bytes of the called probabilities (_bam2modbam.py:206-208): p falls into bin floor(256 p) of [0, 1) cut into 256 bins and
    p >= 1 into the last one; computed on the values as given (Python floats or NumPy float32 scalars)."""
    ml = []
    for p in probs:
        ml.append(255 if not p < 1 else math.floor(p * 256))
    return ml


_PULSE_TAGS = frozenset(("fi", "fp", "ri", "rp"))
_MOD_TAGS = frozenset(("MM", "ML"))


def _refill_tags(all_tags, mm_values, ml_values, rm_pulse=True):
    """The tag list of the output record (_bam2modbam.py:211-226): the input's (tag, value) pairs in order without any earlier
    MM/ML and — unless the kinetics are kept — without fi/fp/ri/rp, followed by MM = 'C+m?,<deltas>;' and ML when the read was
    called.  Value types are left to the writer, as the reference leaves them to pysam."""
    dropped = _MOD_TAGS | _PULSE_TAGS if rm_pulse else _MOD_TAGS
    out = [(t[0], t[1]) for t in all_tags if t[0] not in dropped]
    if mm_values is None:
        return out
    deltas = ",".join(str(v) for v in mm_values)
    return out + [("MM", "C+m?,%s;" % deltas), ("ML", ml_values)]
