"""The NumPy and C oracles against the same model assembled from STOCK PyTorch modules (nn.GRU, matmuls, softmax) — a third,
independent statement of models.py:89-150 / utils/attention.py:48-70 next to the reference-generated goldens.  CPU only."""
import numpy as np
import pytest
import torch

from ccsmeth_amd.utils import synth
from oracle import attbigru2s_oracle as orc


def torch_modules_forward(w, s, h1, h2):
    tw = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in w.items()}
    gru = torch.nn.GRU(11, 256, 3, batch_first=True, bidirectional=True)
    gru.load_state_dict({k[4:]: v for k, v in tw.items() if k.startswith("rnn.")})
    gru.eval()

    def strand(kmer, ipd, pw, npass, h0):
        x = torch.cat([tw["embed.weight"][kmer.long()], ipd[..., None], pw[..., None], npass[:, None, None].expand(-1, 21, 1)], 2)
        out, hn = gru(x, h0)
        q = torch.cat([hn[-2], hn[-1]], 1) @ tw["_att3.Wa.weight"].T
        e = torch.tanh(q[:, None, :] + out @ tw["_att3.Ua.weight"].T) @ tw["_att3.va.weight"].T
        return (torch.softmax(e, 1) * out).sum(1)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in s.items()}
    with torch.no_grad():
        c = torch.cat([strand(t["kmer1"], t["ipd1"], t["pw1"], t["npass1"], torch.from_numpy(h1)),
                       strand(t["kmer2"], t["ipd2"], t["pw2"], t["npass2"], torch.from_numpy(h2))], 1)
        logits = c @ tw["fc1.weight"].T + tw["fc1.bias"]
        return logits.numpy(), torch.softmax(logits, 1).numpy()


@pytest.mark.parametrize("n,seed", [(1, 3), (37, 4), (96, 5)])
def test_oracles_match_stock_torch_modules(n, seed):
    w = synth.synth_weights(seed)
    s = synth.synth_sites(n, seed + 10)
    h1, h2 = synth.synth_h0(n, seed + 20)
    logits, probs = torch_modules_forward(w, s, h1, h2)
    args = (w, s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h1, h2)
    lo, po = orc.attbigru2s_forward(*args)
    assert np.abs(po - probs).max() < 2e-6 and np.abs(lo - logits).max() < 1e-5
    try:
        from oracle import c_oracle
        c_oracle.load()
    except ImportError:
        pytest.skip("oracle/_build/liboracle.so not built")
    lc, pc = c_oracle.forward(*args)
    assert np.abs(pc - probs).max() < 5e-6 and np.abs(lc - logits).max() < 2e-5
