"""Small tables and sequence helpers of the hot path — host mirror of reference ccsmeth/utils/process_utils.py
(:12-15 basepairs, :26-29 base2code_dna, :64-73 constants, :106-118 complement_seq, :426-449 codecv1_to_frame2)."""
import numpy as np

N_VOCAB = 5
NEMBED_BASE = 8

base2code_dna = {'A': 0, 'C': 1, 'G': 2, 'T': 3, 'N': 4, 'W': 4, 'S': 4, 'M': 4, 'K': 4, 'R': 4,
                 'Y': 4, 'B': 4, 'V': 4, 'D': 4, 'H': 4, 'Z': 4}

basepairs = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A', 'N': 'N', 'W': 'W', 'S': 'S', 'M': 'K', 'K': 'M', 'R': 'Y',
             'Y': 'R', 'B': 'V', 'V': 'B', 'D': 'H', 'H': 'D', 'Z': 'Z'}

# 256-entry byte tables (vectorised forms of the two dicts; bases outside the tables: code -> KeyError in the reference's
# batching, complement -> 'N')
_CODE_LUT = np.full(256, 255, dtype=np.uint8)
for _b, _c in base2code_dna.items():
    _CODE_LUT[ord(_b)] = _c
_COMP_LUT = np.full(256, ord('N'), dtype=np.uint8)
for _b, _c in basepairs.items():
    _COMP_LUT[ord(_b)] = ord(_c)


def codecv1_to_frame2():
    """PacBio CodecV1: codes 0-63 -> frames 0-63, 64-127 -> 64+2k, 128-191 -> 192+4k, 192-255 -> 448+8k (max 952)."""
    i = np.arange(64)
    return np.concatenate([i, 64 + 2 * i, 192 + 4 * i, 448 + 8 * i]).tolist()


CODE2FRAMES = np.asarray(codecv1_to_frame2(), dtype=np.int64)


def complement_seq(base_seq, seq_type="DNA"):
    """Reverse complement (IUPAC pair table; unknown letters -> 'N')."""
    if seq_type != "DNA":
        raise ValueError("the seq_type must be DNA (RNA is outside this build)")
    b = np.frombuffer(base_seq.encode("ascii"), dtype=np.uint8)
    return _COMP_LUT[b[::-1]].tobytes().decode("ascii")


def seq_to_codes(ascii_bytes):
    """uint8 ASCII array -> base codes (A0 C1 G2 T3 other-IUPAC 4); raises KeyError like the reference's dict lookup on a
    letter outside base2code_dna."""
    codes = _CODE_LUT[ascii_bytes]
    if (codes == 255).any():
        bad = ascii_bytes[codes == 255][0]
        raise KeyError(chr(int(bad)))
    return codes


def str2bool(v):
    return v.lower() in ("yes", "true", "t", "1")
