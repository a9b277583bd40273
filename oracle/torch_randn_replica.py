"""Bit-level restatement of torch's CPU `torch.randn` stream after `torch.manual_seed(seed)` — TEST INFRASTRUCTURE ONLY
(the product has its own copy of this arithmetic in ccsmeth_amd/aggregate.py; this one pins it against torch draws
captured from the reference run, tests/golden/aggr_golden.npz).

Why it exists: the aggregate-mode caller re-seeds per region (call_mods_freq_bam.py:313 torch.manual_seed(args.tseed)),
so unlike call_mods its h0 = torch.randn(2, B, 32) (models.py:661-671) IS deterministic and parity needs the same numbers.

Published algorithm (ATen, pinned torch<=2.1.0 in the reference; the container's torch 2.10 behaves identically):
  * generator: mt19937, `manual_seed(s)` = init_genrand(s & 0xffffffff)  (ATen/core/MT19937RNGEngine.h)
  * float uniform in [0,1): (x & (2^24 - 1)) * 2^-24 from one 32-bit draw   (ATen/core/TransformationHelper.h:85-88)
  * normal_ on a contiguous CPU float tensor with numel >= 16: fill with uniforms, then per block of 16:
      u1 = 1 - d[j], u2 = d[j+8], r = sqrt(-2 ln u1), theta = 2 pi u2, d[j] = r cos(theta), d[j+8] = r sin(theta)
    (ATen/native/cpu/DistributionTemplates.h:140-149, 208-229); numel % 16 != 0 would redraw the last 16 (not the case for
    (2, B, 32) tensors: numel = 64 B).
  * the reference seeds BEFORE constructing AggrAttRNN (call_mods_freq_bam.py:313-322), whose parameter initialisation
    consumes a fixed number of 32-bit draws; that offset is measured once against the captured draws (`find_offset`).
"""
import numpy as np


class Mt19937Stream:
    def __init__(self, seed):
        self.bg = np.random.MT19937()
        self.bg._legacy_seeding(int(seed) & 0xFFFFFFFF)     # init_genrand(seed)

    def raw(self, n):
        return self.bg.random_raw(int(n)).astype(np.uint32)


def normals_from_raw(raw):
    """raw: uint32 draws, len multiple of 16 -> float32 normals in torch's block-of-16 Box-Muller order."""
    raw = np.asarray(raw, dtype=np.uint32)
    assert raw.size % 16 == 0
    u = ((raw & np.uint32((1 << 24) - 1)).astype(np.float32) * np.float32(1.0 / (1 << 24))).reshape(-1, 16)
    u1 = np.float32(1.0) - u[:, :8]
    u2 = u[:, 8:]
    radius = np.sqrt(np.float32(-2.0) * np.log(u1))
    theta = np.float32(2.0 * np.pi) * u2
    out = np.empty_like(u)
    out[:, :8] = radius * np.cos(theta)
    out[:, 8:] = radius * np.sin(theta)
    return out.reshape(-1)


def find_offset(seed, first_normals, max_offset=200000):
    """Number of 32-bit draws the reference consumed between manual_seed(seed) and its first randn."""
    first_normals = np.asarray(first_normals, np.float32).reshape(-1)[:16]
    s = Mt19937Stream(seed)
    raw = s.raw(max_offset + 16)
    for off in range(max_offset):
        cand = normals_from_raw(raw[off:off + 16])
        if np.allclose(cand, first_normals, rtol=0, atol=2e-6):
            return off
    raise ValueError("offset not found")
