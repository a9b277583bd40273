"""Training step (libccsm_train) on the GPU against the reference model under torch autograd
(tests/golden/make_train_golden.py): loss, logits, every parameter's gradient, and the parameters after three
clip_grad_norm_(0.5) + Adam steps.  fp32 on both sides: the tolerances are those of a different summation order."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from ccsmeth_amd.utils import synth

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(GOLDEN, "train_golden.npz"))
META = json.load(open(os.path.join(GOLDEN, "train_golden.json")))


def _inputs(case, k):
    n = case["n"]
    sites = synth.synth_sites(n, case["site_seed"] + 1000 * k)
    h1, h2 = synth.synth_h0(n, case["h0_seed"] + 1000 * k)
    labels = np.random.default_rng(case["label_seed"] + 1000 * k).integers(0, 2, n).astype(np.int64)
    return sites, (h1, h2), labels


@pytest.mark.parametrize("name", sorted(k for k in META if "steps" in META[k]))
def test_gradients_and_adam_steps_match_reference_autograd(name):
    from ccsmeth_amd.train import Trainer, PARAM_NAMES
    case = META[name]
    assert case["param_names"] == PARAM_NAMES                       # flat order = model.parameters() order of the reference
    tr = Trainer(synth.synth_weights(case["weight_seed"]), device=0, max_sites=case["n"])
    assert tr.num_params == 3043114
    losses, norms = [], []
    for k in range(case["steps"]):
        sites, h0, labels = _inputs(case, k)
        loss, logits = tr.forward_backward(sites, labels, h0=h0, pos_weight=case["pos_weight"], want_logits=True)
        if k == 0:
            assert np.abs(logits - G[name + "_logits"]).max() < 2e-5
            grads = tr.grads()
            for pn in PARAM_NAMES:
                g = grads[pn].ravel()
                ref_norm = float(G["%s_gnorm_%s" % (name, pn)])
                assert abs(np.linalg.norm(g.astype(np.float64)) - ref_norm) <= 2e-4 * ref_norm + 1e-7, pn
                ref = G["%s_g_%s" % (name, pn)]
                got = g[synth.sample_index(pn, g.size, case["sample"])]
                assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-7, (pn, np.abs(got - ref).max(), np.abs(ref).max())
        losses.append(loss)
        norms.append(tr.step(case["lr"], max_norm=0.5))
    assert np.allclose(losses, case["losses"], rtol=2e-3, atol=2e-4), (losses, case["losses"])
    assert np.allclose(norms, case["grad_norms"], rtol=5e-3), (norms, case["grad_norms"])
    sd = tr.state_dict()
    bad = tot = 0
    for pn in PARAM_NAMES:
        v = sd[pn].ravel()
        got = v[synth.sample_index(pn, v.size, case["sample"])]
        ref = G["%s_p_%s" % (name, pn)]
        d = np.abs(got - ref)
        # Adam's first steps move every entry by about lr * sign(g): an entry whose gradient is at the rounding noise
        # can take the step the other way (2 * lr per step); everything else agrees to fp32 accuracy
        assert d.max() <= 2.05 * case["lr"] * case["steps"], (pn, d.max())
        bad += int((d > 2e-5).sum())
        tot += d.size
    assert bad <= 0.01 * tot, (bad, tot)
    tr.close()


def test_eval_matches_inference_library_and_errors():
    """Forward-only path of the trainer vs libccsm (the product's inference kernels) on the same weights and h0."""
    from ccsmeth_amd import _lib
    from ccsmeth_amd.models import DeviceModel
    from ccsmeth_amd.train import Trainer
    w = synth.synth_weights(31)
    n = 130
    s = synth.synth_sites(n, 32)
    h1, h2 = synth.synth_h0(n, 33)
    labels = np.random.default_rng(34).integers(0, 2, n)
    tr = Trainer(w, device=0, max_sites=160)
    loss, logits = tr.evaluate(s, labels, h0=(h1, h2), pos_weight=1.7)
    dm = DeviceModel(w, device=0)
    ws = dm.workspace(n)
    ref_logits, probs = ws.forward_host(s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h0=(h1, h2))
    assert np.abs(logits - ref_logits).max() < 1e-4
    wts = np.where(labels == 1, 1.7, 1.0)
    ref_loss = float((wts * -np.log(probs[np.arange(n), labels])).sum() / wts.sum())
    assert abs(loss - ref_loss) < 1e-4
    dm.close()
    with pytest.raises(_lib.CcsmError) as e:
        tr.forward_backward(synth.synth_sites(161, 1), np.zeros(161, np.int64), h0="zero")
    assert e.value.status == _lib.ERR_CAPACITY
    with pytest.raises(_lib.CcsmError):
        tr.forward_backward(s, labels, h0=(h1, h2), dropout_rate=1.0)
    with pytest.raises(ValueError):
        tr.forward_backward(s, labels[:-1], h0=(h1, h2))
    tr.close()


def test_dropout_and_device_rng_h0_train_and_reduce_loss():
    """With dropout 0.5 and device-drawn h0 the model learns a synthetic labelling (label = sign of the centre IPDs) in a
    few hundred steps; the same seed reproduces the same masks and h0; dropout 0 differs from dropout 0.5."""
    from ccsmeth_amd.train import Trainer
    w = synth.synth_weights(41)
    n = 512
    pool = synth.synth_sites(n * 8, 42)
    val = synth.synth_sites(n, 43)
    lab = lambda s: (s["ipd1"][:, 10] + s["ipd2"][:, 10] > 0).astype(np.int64)  # noqa: E731

    def run(rate, seed, steps):
        tr = Trainer(w, device=0, max_sites=n)
        out = []
        for k in range(steps):
            i = (k % 8) * n
            s = {key: v[i:i + n] for key, v in pool.items()}
            loss, _ = tr.forward_backward(s, lab(s), h0=None, dropout_rate=rate, seed=seed, step=k)
            tr.step(1e-3)
            out.append(loss)
        ev, logits = tr.evaluate(val, lab(val), h0=None, seed=seed, step=10 ** 6)
        acc = float((logits.argmax(1) == lab(val)).mean())
        trained.append(tr.state_dict())
        tr.close()
        return out, ev, acc
    trained = []
    a, eva, acca = run(0.5, 7, 320)
    b, _, _ = run(0.5, 7, 4)
    c, _, _ = run(0.0, 7, 4)
    assert a[:4] == b                                  # same masks and h0, and no float atomics in the step: the same bits
    assert not np.allclose(a[1:4], c[1:4], rtol=1e-3)
    assert np.mean(a[-10:]) < 0.45 and eva < 0.45 and acca > 0.85, (a[:3], a[-10:], eva, acca)
    # the TRAINED parameters (not the synthetic initialisation the other parity tests use) through the inference library's default:
    # whatever the rule of ccsm_create selects for THIS checkpoint (a trained model is more sensitive: split-mx leaves ~1 % of the
    # sites beyond 1e-5 with a heavy tail, so the probe ends at its first batch and the three-pass arithmetic is served), the
    # probabilities stay within that arithmetic's bound of the oracle (h0 pinned).  The committed checkpoints of
    # tests/test_gpu_zz_trained_checkpoints.py are the deterministic version of this check.
    from ccsmeth_amd.models import DeviceModel
    from oracle import attbigru2s_oracle as orc
    wt = trained[0]
    dm = DeviceModel(wt, device=0)
    ok = dm.probe_sites == 65536 and dm.probe_error <= 1.25e-5 and dm.probe_error <= 3.0 * dm.probe_q999     # (ccsm_create's acceptance rule)
    assert dm.precision == (4 if ok else 3), (dm.precision, dm.probe_error, dm.probe_q999, dm.probe_sites)
    m = 256
    sv = {k: v[:m] for k, v in val.items()}
    h1, h2 = synth.synth_h0(m, 99)
    ws = dm.workspace(m)
    _, probs = ws.forward_host(sv["kmer1"], sv["ipd1"], sv["pw1"], sv["npass1"], sv["kmer2"], sv["ipd2"], sv["pw2"], sv["npass2"], h0=(h1, h2))
    ws.close(); dm.close()
    _, ref = orc.attbigru2s_forward(wt, sv["kmer1"], sv["ipd1"], sv["pw1"], sv["npass1"], sv["kmer2"], sv["ipd2"], sv["pw2"], sv["npass2"], h1, h2)
    assert np.abs(probs - ref).max() < (2e-5 if dm.precision >= 4 else 2e-6), dm.precision
    assert 0.05 < float((ref[:, 1] > 0.5).mean()) < 0.95                # a model that actually discriminates


def test_large_gate_gradients_fall_back_to_the_stepwise_backward(monkeypatch):
    """The fused backward kernel's scaled-fp16 operand saturates above |gate gradient| = 14.6.  The loss is a weighted MEAN, so ordinary
    models stay orders of magnitude below that; an fc1 layer blown up by 3e6 gets there.  The step is then flagged on the device and
    its backward pass repeated step by step on the same saved activations, so the gradients equal those of a trainer whose backward
    pass is stepwise from the start (CCSM_TRAIN_STEPWISE=bwd: same fused forward, hence the same loss gradient - a trainer that is
    stepwise throughout computes a forward that differs in the last bits, which this ill-conditioned model turns into other
    misclassified sites); ordinary gradients never take the detour."""
    from ccsmeth_amd.train import Trainer
    n = 512                                              # 1024 strand rows: the fused path
    sites = synth.synth_sites(n, 61)
    labels = (np.arange(n) % 97 == 0).astype(np.int64)   # a few positives carry the whole weighted loss
    h1, h2 = synth.synth_h0(n, 62)
    w = synth.synth_weights(9)
    big = dict(w)
    big["fc1.weight"] = (w["fc1.weight"] * 3e6).astype(np.float32)
    out = {}
    for mode in ("fused", "stepwise"):
        if mode == "stepwise":
            monkeypatch.setenv("CCSM_TRAIN_STEPWISE", "bwd")
        tr = Trainer(w, device=0, max_sites=n)
        l_small, _ = tr.forward_backward(sites, labels, h0=(h1, h2), pos_weight=2.0)
        g_small = tr.grads()
        fb0 = tr.fused_fallbacks
        tr.close()
        tr = Trainer(big, device=0, max_sites=n)
        l_big, _ = tr.forward_backward(sites, labels, h0=(h1, h2), pos_weight=2.0)
        out[mode] = (g_small, tr.grads(), fb0, tr.fused_fallbacks, l_small, l_big)
        tr.close()
    assert out["fused"][2] == 0 and out["stepwise"][3] == 0
    if out["fused"][3] == 0:
        pytest.skip("fc1 x 3e6 did not saturate the fused kernel on this initialisation")
    for k in out["fused"][1]:
        a, b = out["fused"][1][k], out["stepwise"][1][k]
        assert np.allclose(a, b, rtol=1e-4, atol=1e-6 * np.abs(b).max()), k


def test_device_drawn_initial_states_advance_by_sites():
    """The h0 generator's counter is the running SITE index (trainm: step * batch_size): site j of a batch drawn at offset o + 1
    gets the window site j + 1 gets at offset o, and two consecutive steps of N sites share no window at all (the defect this
    pins: an offset that advanced by 1 per STEP made consecutive steps reuse all but one site's initial states)."""
    from ccsmeth_amd.train import Trainer
    n = 40
    tr = Trainer(synth.synth_weights(3), device=0, max_sites=n)
    sites = synth.synth_sites(n, 12)
    shifted = {k: v[1:] for k, v in sites.items()}
    _, l0 = tr.evaluate(sites, h0=None, seed=5, h0_offset=1000)
    _, l1 = tr.evaluate(shifted, h0=None, seed=5, h0_offset=1001)
    _, l2 = tr.evaluate(shifted, h0=None, seed=5, h0_offset=1000)
    assert np.allclose(l0[1:], l1, atol=2e-5)                           # same site, same counter window -> the same logits
    assert not np.isclose(l0[1:], l2, atol=1e-4).all(axis=1).any()      # ... and other windows give other logits
    _, a = tr.evaluate(sites, h0=None, seed=5, step=7)                  # default: offset = step * N
    _, b = tr.evaluate(sites, h0=None, seed=5, step=8)
    _, c = tr.evaluate(sites, h0=None, seed=5, h0_offset=8 * n)
    assert np.array_equal(b, c)
    assert not np.isclose(a, b, atol=1e-6).all(axis=1).any()             # no site sees the same initial states in consecutive steps
    for shift in range(1, n):                                           # ... nor a neighbour's
        assert not np.isclose(a[shift:], b[:n - shift], atol=1e-7).all(axis=1).any()


def _write_features(path, sites, labels):
    code2base = "ACGTN"
    with open(path, "w") as wf:
        for i in range(len(labels)):
            c = lambda a: ",".join(repr(float(x)) for x in a)  # noqa: E731
            wf.write("\t".join([".", "-1", ".", "m/%d/ccs" % i, "100",
                                "".join(code2base[b] for b in sites["kmer1"][i]), str(int(sites["npass1"][i])), c(sites["ipd1"][i]), ".",
                                c(sites["pw1"][i]), ".", ".", ".",
                                "".join(code2base[b] for b in sites["kmer2"][i]), str(int(sites["npass2"][i])), c(sites["ipd2"][i]), ".",
                                c(sites["pw2"][i]), ".", ".", ".", str(int(labels[i]))]) + "\n")


def _make_tables(tmp_path, n_train=4096, n_valid=1024):
    lab = lambda s: (s["ipd1"][:, 10] + s["ipd2"][:, 10] > 0).astype(np.int64)  # noqa: E731
    tr, va = synth.synth_sites(n_train, 61), synth.synth_sites(n_valid, 62)
    _write_features(str(tmp_path / "train.tsv"), tr, lab(tr))
    _write_features(str(tmp_path / "valid.tsv"), va, lab(va))
    return va, lab(va)


def test_trainm_end_to_end_checkpoint_serves_inference(tmp_path):
    """`trainm` from feature tables to checkpoints: epochs run, rank 0 writes <model_type>.b21_epoch<k>.ckpt with the reference's
    state_dict keys, and the best checkpoint loads into the inference library (libccsm) and classifies the validation set."""
    import torch
    from ccsmeth_amd import trainm
    from ccsmeth_amd.models import DeviceModel
    va, vlab = _make_tables(tmp_path)
    (tmp_path / "models").mkdir()
    (tmp_path / "models" / "attbigru2s.b21_epoch99.ckpt").write_text("stale")          # removed at start like the reference does
    args = trainm.build_parser().parse_args(["--train_file", str(tmp_path / "train.tsv"), "--valid_file", str(tmp_path / "valid.tsv"),
                                             "--model_dir", str(tmp_path / "models"), "--max_epoch_num", "30", "--min_epoch_num", "30",
                                             "--lr_decay", "1.0", "--batch_size", "256", "--step_interval", "16"])
    res = trainm.train(args, log=open(os.devnull, "w"))
    assert res["epochs"] == 30 and res["steps"] == 30 * 16 and res["best_acc"] > 0.85, res
    files = sorted(os.listdir(tmp_path / "models"))
    assert "attbigru2s.b21_epoch99.ckpt" not in files and "attbigru2s.b21_epoch%d.ckpt" % res["best_epoch"] in files
    sd = torch.load(str(tmp_path / "models" / ("attbigru2s.b21_epoch%d.ckpt" % res["best_epoch"])), map_location="cpu")
    from ccsmeth_amd.train import PARAM_NAMES
    assert list(sd.keys()) == PARAM_NAMES
    dm = DeviceModel({k: v.numpy() for k, v in sd.items()}, device=0)
    ws = dm.workspace(len(vlab))
    _, probs = ws.forward_host(va["kmer1"], va["ipd1"], va["pw1"], va["npass1"], va["kmer2"], va["ipd2"], va["pw2"], va["npass2"])
    assert float((probs.argmax(1) == vlab).mean()) > 0.8
    dm.close()


def test_trainm_two_ranks_stay_in_sync(tmp_path):
    """Two ranks (sharing the one GPU of the test box, gradients averaged over gloo): both replicas end with the same
    parameters, each ran half of every epoch's steps, and the model learns."""
    import subprocess
    import sys
    from conftest import ROOT
    _make_tables(tmp_path, 4096, 512)
    env = dict(os.environ, CCSM_DIST_BACKEND="gloo", CCSM_TRAINM_REPORT=str(tmp_path / "report"), PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29100 + os.getpid() % 1000), "-m", "ccsmeth_amd", "trainm", "--train_file", str(tmp_path / "train.tsv"),
           "--valid_file", str(tmp_path / "valid.tsv"), "--model_dir", str(tmp_path / "m2"), "--max_epoch_num", "30", "--min_epoch_num", "30",
           "--lr_decay", "1.0", "--batch_size", "128"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert out.returncode == 0, out.stdout.decode()[-3000:]
    r0, r1 = (json.load(open(str(tmp_path / "report") + ".rank%d.json" % r)) for r in (0, 1))
    assert r0["world"] == r1["world"] == 2 and r0["steps"] == r1["steps"] == 30 * 16
    assert abs(r0["param_checksum"] - r1["param_checksum"]) <= 1e-7 * r0["param_checksum"]
    assert r0["best_acc"] > 0.75, (r0, out.stdout.decode()[-2000:])


def test_graph_replay_of_the_step_equals_eager(monkeypatch):
    """CCSM_TRAIN_GRAPH=1: the first call of a batch shape runs eagerly, the second is captured (both streams, rocBLAS calls
    included) and later ones replay the graph; per-step seeds / class weights come from device memory, so the run equals the
    eager one."""
    from ccsmeth_amd.train import Trainer
    w = synth.synth_weights(51)
    n = 192
    batches = [(synth.synth_sites(n, 52 + k), np.random.default_rng(60 + k).integers(0, 2, n)) for k in range(5)]

    def run():
        tr = Trainer(w, device=0, max_sites=n)
        out = []
        for k, (s, lab) in enumerate(batches):
            loss, logits = tr.forward_backward(s, lab, h0=None, pos_weight=1.0 + 0.5 * k, dropout_rate=0.3, seed=9, step=k, want_logits=True)
            out.append((loss, logits, tr.grads()["rnn.weight_hh_l1"].copy(), tr.step(1e-3)))
        ev = tr.evaluate(batches[0][0], batches[0][1], h0=None, seed=9, step=99)
        ev2 = tr.evaluate(batches[1][0], batches[1][1], h0=None, seed=9, step=100)       # replayed eval graph
        tr.close()
        return out, ev, ev2
    monkeypatch.setenv("CCSM_TRAIN_GRAPH", "0")
    eager, e1, e2 = run()
    monkeypatch.setenv("CCSM_TRAIN_GRAPH", "1")
    graph, g1, g2 = run()
    for (la, lo_a, ga, na), (lb, lo_b, gb, nb) in zip(eager[:2], graph[:2]):
        assert abs(la - lb) < 1e-5 and np.abs(lo_a - lo_b).max() < 1e-4 and np.abs(ga - gb).max() <= 1e-4 * np.abs(ga).max() and abs(na - nb) < 1e-3 * na
    assert np.allclose([x[0] for x in eager], [x[0] for x in graph], rtol=2e-2)       # later steps: fp32 reduction order only
    assert abs(e1[0] - g1[0]) < 2e-2 and abs(e2[0] - g2[0]) < 2e-2


def test_whole_chain_bam_extract_trainm_call_mods(tmp_path):
    """The reference's workflow end to end without the reference: HiFi BAM -> `extract` (two label classes) -> `trainm` ->
    checkpoint -> `call_mods` -> modbam whose ML values follow the class the model was trained to separate."""
    from ccsmeth_amd import bamio, extract_cli, trainm
    from ccsmeth_amd.call_mods import build_parser, call_mods
    rng = np.random.default_rng(5)

    def write_bam(path, shift, n_reads=24):
        with bamio.BamWriter(path, "@HD\tVN:1.5\tSO:unknown\n", []) as w:
            for i in range(n_reads):
                L = int(rng.integers(900, 1500))
                seq = rng.choice(list("ACGT"), size=L)
                for j in range(11, L - 12, 19):
                    seq[j], seq[j + 1] = "C", "G"
                kin = lambda s: np.clip(rng.gamma(2.0, 12.0, size=L) + s, 0, 255).astype(np.uint8)  # noqa: E731
                fi, ri = kin(0), kin(0)
                cg = np.flatnonzero((seq[:-1] == "C") & (seq[1:] == "G"))
                fi[cg] = np.clip(fi[cg].astype(int) + shift, 0, 255)                 # the "methylated" class: slower IPD at the C
                w.write(bamio.BamRecord("m/%d/ccs" % (i + 1000 * shift), flag=4, seq="".join(seq),
                                        tags=[("fi", "BC", fi), ("ri", "BC", ri), ("fp", "BC", kin(0)), ("rp", "BC", kin(0)), ("fn", "C", 10), ("rn", "C", 11)]))
    tables = []
    for label, shift in ((0, 0), (1, 60)):
        bam = str(tmp_path / ("c%d.bam" % label))
        write_bam(bam, shift)
        res = extract_cli.extract_hifireads_features(extract_cli.build_parser().parse_args(["-i", bam, "--methy_label", str(label)]), log=open(os.devnull, "w"))
        tables.append(open(res["output"]).read().splitlines(True))
    lines = tables[0] + tables[1]
    order = rng.permutation(len(lines))
    cut = int(0.85 * len(lines))
    open(str(tmp_path / "train.tsv"), "w").writelines(lines[i] for i in order[:cut])
    open(str(tmp_path / "valid.tsv"), "w").writelines(lines[i] for i in order[cut:])
    res = trainm.train(trainm.build_parser().parse_args(["--train_file", str(tmp_path / "train.tsv"), "--valid_file", str(tmp_path / "valid.tsv"),
                                                         "--model_dir", str(tmp_path / "m"), "--max_epoch_num", "25", "--min_epoch_num", "25",
                                                         "--lr_decay", "1.0", "--batch_size", "256"]), log=open(os.devnull, "w"))
    assert res["best_acc"] > 0.85, res
    ckpt = str(tmp_path / "m" / ("attbigru2s.b21_epoch%d.ckpt" % res["best_epoch"]))
    means = []
    for label in (0, 1):
        out = call_mods(build_parser().parse_args(["-i", str(tmp_path / ("c%d.bam" % label)), "-m", ckpt, "-o", str(tmp_path / ("o%d" % label))]),
                        log=open(os.devnull, "w"))
        with bamio.BamReader(out["output"]) as rd:
            ml = np.concatenate([r.get_tag("ML") for r in rd if r.has_tag("ML")])
        means.append(float(ml.mean()))
    assert means[1] > means[0] + 100, means                     # ML bytes: the trained model separates the two classes


@pytest.mark.parametrize("t_a,t_b", [(False, True), (False, False), (True, False), (True, True)])
def test_training_gemm_kernel_against_float64(t_a, t_b):
    """The matrix-product kernel of the training step (ccsm_train_gemm.hip: three-pass split-fp16 MFMA, fp32 in / out) in its four storage
    forms against NumPy float64: ragged shapes (K = 11 as in layer 0; M, N not multiples of the 128 x 128 tile), padded row strides,
    alpha / beta, and a GRADIENT-sized operand (1e-7: far below fp16's normal range; scaled by its maximum inside the library).
    Tolerance: 2^-20 of the product of the operands' magnitudes times sqrt(K) - fp32-class."""
    from ccsmeth_amd.train import selftest_gemm
    rng = np.random.default_rng(12)
    for (M, N, K, pad, mag, grad) in ((200, 768, 11, 0, 1.0, False), (257, 130, 513, 3, 1.0, False), (96, 40, 2100, 0, 1.0, False),
                                      (300, 512, 768, 0, 1e-7, True), (64, 11, 8400, 5, 3e-6, True)):
        a_log = (rng.standard_normal((M, K)) * mag).astype(np.float32)          # op(A): M x K
        b_log = (rng.standard_normal((K, N)) * 0.1).astype(np.float32)          # op(B): K x N
        a_st = np.ascontiguousarray(a_log.T if t_a else a_log)
        b_st = np.ascontiguousarray(b_log.T if t_b else b_log)
        if pad:                                                                  # views with a row stride larger than the row
            big = np.zeros((a_st.shape[0], a_st.shape[1] + pad), np.float32); big[:, :a_st.shape[1]] = a_st; a_st = big[:, :a_st.shape[1]]
            big = np.zeros((b_st.shape[0], b_st.shape[1] + pad), np.float32); big[:, :b_st.shape[1]] = b_st; b_st = big[:, :b_st.shape[1]]
        c0 = rng.standard_normal((M, N)).astype(np.float32) * mag
        got = selftest_gemm(a_st, b_st, c0, t_a=t_a, t_b=t_b, alpha=0.5, beta=2.0, grad_a=grad)
        want = 0.5 * (a_log.astype(np.float64) @ b_log.astype(np.float64)) + 2.0 * c0.astype(np.float64)
        tol = 2.0 ** -20 * mag * 0.1 * np.sqrt(K) * 4 + 2.0 ** -22 * np.abs(want).max()
        assert np.abs(got - want).max() <= tol, (t_a, t_b, M, N, K, float(np.abs(got - want).max()), tol)


@pytest.mark.parametrize("n", [130, 512])
def test_training_is_bit_reproducible(n):
    """Two trainings from the same seed end with the SAME BITS - losses, gradient norms and every parameter - on the stepwise path (130
    sites) and the fused one (512): since round 4 no reduction of the step uses float atomics (per-block partial sums added up in a fixed
    order; the matrix products, their gradient-operand scales and the gradient norm are order-fixed too).  What a committed checkpoint
    recipe needs to mean something."""
    from ccsmeth_amd.train import Trainer, PARAM_NAMES
    w = synth.synth_weights(23)
    pool, labels = synth.synth_labeled_sites(n * 4, 77)

    def run():
        tr = Trainer(w, device=0, max_sites=n)
        out = []
        for k in range(12):
            i = (k % 4) * n
            q = {key: v[i:i + n] for key, v in pool.items()}
            loss, _ = tr.forward_backward(q, labels[i:i + n], h0=None, pos_weight=1.3, dropout_rate=0.5, seed=9, step=k)
            out.append((loss, tr.step(1e-3)))
        sd = tr.state_dict()
        tr.close()
        return out, sd
    a, pa = run()
    b, pb = run()
    assert a == b, (a[:3], b[:3])
    for k in PARAM_NAMES:
        assert np.array_equal(pa[k], pb[k]), k
