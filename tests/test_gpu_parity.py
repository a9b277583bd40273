"""GPU parity tests proper: the HIP path (through the C-ABI) against the NumPy oracle on the same seeded inputs,
against the committed reference goldens, and size-independent properties at BASELINE.json's batch size."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from ccsmeth_amd.utils import synth

pytestmark = pytest.mark.gpu

PROB_TOL = 1e-4      # BASELINE.json north_star: per-site probabilities within 1e-4 of the reference
LOGIT_TOL = 5e-4


@pytest.fixture(scope="module")
def lib():
    from ccsmeth_amd import _lib
    return _lib


DEFAULT_TOL = 1e-5   # what the default (split-mx) arithmetic has to hold on well-behaved checkpoints (measured: 4-5e-6)
SPLIT3_TOL = 1e-6    # the fp32-class fallback (measured: 2-3e-7)


@pytest.fixture(scope="module", params=[0, 6, 5, 3], ids=["default-split-mx", "split-mx-d", "hybrid", "split3"])
def model7(request):
    """Every test on this fixture runs in the default arithmetic (split-mx, CCSM_PRECISION_SPLIT_F8), in split-mx-d (fp6 recurrent weights,
    per-row state scales) and the hybrid (split-mx input part, three-pass recurrent part) - what the probe selects for trained
    checkpoints - and in the three-pass fp16 split."""
    from ccsmeth_amd.models import DeviceModel
    w = synth.synth_weights(7)
    dm = DeviceModel(w, device=0, precision=request.param)
    assert dm.precision == {0: 4, 6: 6, 5: 5, 3: 3}[request.param]
    yield w, dm
    dm.close()


def _oracle(w, s, h1, h2):
    from oracle import attbigru2s_oracle as orc
    return orc.attbigru2s_forward(w, s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"],
                                  s["npass2"], h1, h2)


def _fwd(ws, s, h0, **kw):
    return ws.forward_host(s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"],
                           h0=h0, **kw)


def test_mfma_fragment_convention(lib):
    err = C.c_float(1.0)
    lib.check(lib.load().ccsm_selftest_mfma(0, C.byref(err)))
    assert err.value < 1e-3


def test_mfma_ceiling_probe_modes(lib):
    """ccsm_measure_mfma_ceiling: what the matrix cores of this device sustain under its power cap on register-resident random operands, by
    instruction: 0 / 3 = fp16 on 32x32x16 / 16x16x32, 1 / 4 = the split-mx mix on the 32- / 16-wide instructions, 2 = the 32-wide mix fed from
    LDS.  Short runs (the figures bench.py quotes take 2-3 s per mode): every mode returns a rate between a tenth of and the datasheet peak, the
    mixes lie below their fp16 instruction, and an unknown mode is refused."""
    l = lib.load()
    tf = {}
    for mode in (0, 1, 2, 3, 4):
        v, g = C.c_float(0), C.c_float(0)
        lib.check(l.ccsm_measure_mfma_ceiling(0, mode, 0.4, C.byref(v), C.byref(g)))
        tf[mode] = v.value
        assert 250.0 < v.value < 2600.0 and g.value > 0.1, (mode, v.value, g.value)
    assert tf[1] < tf[0] and tf[4] < tf[3] and tf[2] <= tf[1] * 1.05, tf
    v = C.c_float(0)
    assert l.ccsm_measure_mfma_ceiling(0, 5, 0.4, C.byref(v), None) != 0


def test_split_f8_product_selftest(lib):
    """One 32x32x32 product in SPLIT_F8 arithmetic (attention pool): the fp8 correction MFMA must remove most of the fp16 operand error."""
    a, b = C.c_float(1.0), C.c_float(0.0)
    lib.check(lib.load().ccsm_selftest_split_f8(0, C.byref(a), C.byref(b)))
    assert a.value < 2e-5 and a.value < b.value / 8


@pytest.mark.parametrize("fmt,bound", [(2, 1e-5), (4, 3e-5)], ids=["fp6-weight-blob", "fp4-weight-blob"])
def test_split_mx_product_selftest(lib, fmt, bound):
    """One pair product of the GRU layers' split-mx arithmetic: host-packed weight blob (fp6 = input part, fp4 = recurrent part) x
    device-packed fp6 activation blob; the host's fp6 encoder must produce the instruction's bytes."""
    a, b, m = C.c_float(1.0), C.c_float(0.0), C.c_int(-1)
    lib.check(lib.load().ccsm_selftest_split_mx(0, fmt, C.byref(a), C.byref(b), C.byref(m)))
    assert m.value == 0
    assert a.value < bound and a.value < b.value / 4


def test_default_arithmetic_within_1e5_and_split3_within_1e6(model7):
    """The tolerances the two arithmetics are held to on the synthetic checkpoint, 2048 sites (the bar of BASELINE.json is 1e-4)."""
    w, dm = model7
    n = 2048
    s = synth.synth_sites(n, 4711)
    h1, h2 = synth.synth_h0(n, 4712)
    ws = dm.workspace(n)
    _, probs = _fwd(ws, s, (h1, h2))
    ws.close()
    err = np.abs(probs - _oracle(w, s, h1, h2)[1]).max()
    assert err < (DEFAULT_TOL if dm.precision >= 4 else SPLIT3_TOL), err


def _cascade(dm):
    """What ccsm_create's probe has to select (include/ccsm.h): split-mx iff all 65536 probe sites stay within 1.25e-5 of the three-pass
    arithmetic and the maximum within 3 x the 99.9th percentile (a light tail); the three-pass arithmetic otherwise.  split-mx-d and the
    hybrid are never probed."""
    assert dm.probe_error_hybrid < 0 and dm.probe_tail_hybrid < 0 and dm.probe_error_mxd < 0
    ok = dm.probe_sites == 65536 and 0 <= dm.probe_error <= 1.25e-5 and dm.probe_error <= 3.0 * dm.probe_q999
    if not ok:
        assert dm.probe_error > 1.25e-5 or dm.probe_sites == 65536     # the probe ends early only on the first condition
    return 4 if ok else 3


def test_default_arithmetic_is_chosen_by_the_probe():
    """precision 0: ccsm_create measures split-mx against split-fp16 on 65536 probe sites; the synthetic checkpoint keeps split-mx, the
    hostile one (Student-t matrices with x50 outliers, gate-saturating biases) is served in whatever arithmetic the rule leaves, and
    in either case holds the tolerance that arithmetic promises.  An explicit precision is never overridden and runs no probe."""
    from ccsmeth_amd.models import DeviceModel
    n = 512
    s = synth.synth_sites(n, 99)
    h1, h2 = synth.synth_h0(n, 98)
    w = synth.synth_weights(7)
    dm = DeviceModel(w, device=0)
    assert dm.precision == 4 and 0 <= dm.probe_error <= 1.25e-5 and dm.probe_tail == 0 and dm.probe_sites == 65536 and 0 < dm.quant_error < 0.2
    dm.close()
    for seed in (7, 11):
        wh = synth.synth_weights_heavy(seed)
        ref = _oracle(wh, s, h1, h2)[1]
        dm = DeviceModel(wh, device=0)
        assert dm.probe_error >= 0
        ws = dm.workspace(n)
        err = np.abs(_fwd(ws, s, (h1, h2))[1] - ref).max()
        assert dm.precision == _cascade(dm), (dm.precision, dm.probe_error, dm.probe_error_hybrid)
        assert err < (2e-5 if dm.precision >= 4 else SPLIT3_TOL), (seed, dm.precision, dm.probe_error, err)
        dm.close()
        forced = DeviceModel(wh, device=0, precision=4)       # explicit split-mx on the hostile checkpoint: still inside the bar
        assert forced.precision == 4 and forced.probe_error < 0 and forced.probe_sites == 0
        ws = forced.workspace(n)
        err4 = np.abs(_fwd(ws, s, (h1, h2))[1] - ref).max()
        forced.close()
        assert err4 < PROB_TOL, (seed, err4)


@pytest.mark.parametrize("scale", [3.0, 8.0])
def test_large_initial_states(model7, scale):
    """|h0| far outside (-1, 1) (|h_t| <= max(1, |h0|): the state stays large).  Through the host-pointer entry points a call whose
    explicit initial states exceed split-mx's domain (|h0| > 6) is served in the split-fp16 arithmetic: fp32-class in both models."""
    w, dm = model7
    n = 200
    s = synth.synth_sites(n, 555)
    h1, h2 = synth.synth_h0(n, 556)
    h1, h2 = (h1 * scale).astype(np.float32), (h2 * scale).astype(np.float32)
    assert np.abs(h1).max() > 6
    ws = dm.workspace(n)
    _, probs = _fwd(ws, s, (h1, h2))
    _, again = _fwd(ws, s, (np.clip(h1, -1, 1), np.clip(h2, -1, 1)))          # the next call is back in the model's own arithmetic
    ws.close()
    assert np.abs(probs - _oracle(w, s, h1, h2)[1]).max() < 1e-5      # fp32 rounding of states up to 35 (measured: 3e-6)
    assert np.abs(again - _oracle(w, s, np.clip(h1, -1, 1), np.clip(h2, -1, 1))[1]).max() < (DEFAULT_TOL if dm.precision >= 4 else SPLIT3_TOL)


def test_large_device_resident_initial_states_degrade_gracefully():
    """Device-resident initial states are not inspected: split-mx then saturates its correction operands (never a NaN: the fp8 / fp6
    conversions are fed clamped values) and stays inside the bar up to |h0| ~ 13."""
    import torch
    from ccsmeth_amd.models import DeviceModel
    w = synth.synth_weights(7)
    n = 200
    s = synth.synth_sites(n, 555)
    h1, h2 = synth.synth_h0(n, 556)
    dm = DeviceModel(w, device=0, precision=4)
    ws = dm.workspace(n)
    dev = torch.device("cuda:0")
    arrs = [torch.from_numpy(np.ascontiguousarray(s[k])).to(dev) for k in ("kmer1", "ipd1", "pw1", "npass1", "kmer2", "ipd2", "pw2", "npass2")]
    for scale, tol in ((3.0, 5e-5), (8.0, None)):
        a, b = (h1 * scale).astype(np.float32), (h2 * scale).astype(np.float32)
        _, probs = ws.forward_torch(*arrs, h0=(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)))
        probs = probs.cpu().numpy()
        assert np.isfinite(probs).all()
        if tol is not None:
            assert np.abs(probs - _oracle(w, s, a, b)[1]).max() < tol
    dm.close()


@pytest.mark.parametrize("n", [1, 31, 32, 33, 100, 513])
def test_forward_vs_oracle_ragged_sizes(model7, n):
    w, dm = model7
    s = synth.synth_sites(n, 1000 + n)
    h1, h2 = synth.synth_h0(n, 2000 + n)
    ws = dm.workspace(n)
    logits, probs = _fwd(ws, s, (h1, h2))
    rl, rp = _oracle(w, s, h1, h2)
    ws.close()
    assert np.isfinite(logits).all() and np.isfinite(probs).all()
    assert np.abs(probs - rp).max() < (DEFAULT_TOL if dm.precision >= 4 else SPLIT3_TOL)
    assert np.abs(logits - rl).max() < LOGIT_TOL


def test_forward_vs_reference_goldens():
    """Committed outputs of the reference itself (tests/golden/make_golden.py), full b21 model."""
    from ccsmeth_amd.models import DeviceModel
    fwd = np.load(os.path.join(GOLDEN, "forward_golden.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "forward_golden.json")))
    for name in ("b21_n1", "b21_n64", "b21_n513", "b21_n2048"):
        m = meta[name]
        w = synth.synth_weights(m["weight_seed"])
        s = synth.synth_sites(m["n"], m["site_seed"])
        h1, h2 = synth.synth_h0(m["n"], m["h0_seed"])
        for prec in (0, 5, 3):
            dm = DeviceModel(w, device=0, precision=prec)
            ws = dm.workspace(m["n"])
            logits, probs = _fwd(ws, s, (h1, h2))
            dm.close()
            assert np.abs(probs - fwd[name + "_probs"]).max() < PROB_TOL, (name, prec)
            assert np.abs(logits - fwd[name + "_logits"]).max() < LOGIT_TOL, (name, prec)


@pytest.mark.parametrize("prec", [0, 6, 5, 3])
def test_model_variants_vs_reference_goldens(prec):
    """is_stds / is_sn / is_map / no is_npass (models.py:39-47, 100-123) against the reference's own outputs, through the raw workspace and
    through ModelAttRNN.forward's 16-argument signature.  The 17- and 18-column combinations (is_stds + is_sn + one or two more) do not
    fit the layer-0 kernels' 16-wide k-block as [embedding(8) | features]: ccsm_create folds the embedding table into the matrix and feeds
    [one-hot(5) | features] = 14 or 15 columns - the same product - so they run in every arithmetic like the others."""
    from ccsmeth_amd.models import DeviceModel, ModelAttRNN
    from test_oracle_golden import VAR, VAR_META, variant_inputs
    wide = 0
    for name, meta in sorted(VAR_META.items()):
        w, s, h1, h2, ex, feats = variant_inputs(meta)
        kw = dict(is_npass=feats[0], is_stds=feats[1], is_sn=feats[2], is_map=feats[3])
        wide += 8 + synth.feas_ccs_of(*feats) > 16
        dm = DeviceModel(w, device=0, precision=prec, **kw)
        ws = dm.workspace(meta["n"])
        logits, probs = ws.forward_host(s["kmer1"], s["ipd1"], s["pw1"], s["npass1"] if feats[0] else None, s["kmer2"], s["ipd2"], s["pw2"],
                                        s["npass2"] if feats[0] else None, h0=(h1, h2), extra=ex)
        if any(feats[1:]):
            with pytest.raises(ValueError):                          # the planes are required
                _fwd(ws, s, (h1, h2))
        ws.close(); dm.close()
        assert np.abs(probs - VAR[name + "_probs"]).max() < (DEFAULT_TOL if prec != 3 else 2e-6), (name, prec)
        assert np.abs(logits - VAR[name + "_logits"]).max() < LOGIT_TOL, (name, prec)
        if prec == 0:
            m = ModelAttRNN(21, 3, 2, 0, 256, model_type="attbigru2s", device=0, **kw)
            m.load_state_dict(w)
            z = np.zeros(meta["n"], np.float32)
            g = lambda d, k: d.get(k, z)      # noqa: E731  (placeholders where a flag is off, as the reference's callers pass)
            rep = lambda a: np.repeat(np.asarray(a, np.float32)[:, None], 21, axis=1)     # noqa: E731
            _, p2 = m(s["kmer1"].astype(np.float32), rep(s["npass1"]), s["ipd1"], g(ex[0], "ipd_std"), s["pw1"], g(ex[0], "pw_std"),
                      g(ex[0], "sn"), g(ex[0], "map"),
                      s["kmer2"].astype(np.float32), rep(s["npass2"]), s["ipd2"], g(ex[1], "ipd_std"), s["pw2"], g(ex[1], "pw_std"),
                      g(ex[1], "sn"), g(ex[1], "map"), h0=(h1, h2))
            m._release()
            assert np.abs(np.asarray(p2) - VAR[name + "_probs"]).max() < DEFAULT_TOL, name
    assert wide == 3                                                 # 17, 17 and 18 columns
    with pytest.raises(ValueError, match="columns"):                 # matrix and flags must agree
        DeviceModel(synth.synth_weights(1, feas_ccs=9), device=0, is_npass=True, is_stds=True, is_sn=False)


def test_workgroup_forms_agree_bitwise(model7, monkeypatch):
    """Launches that would leave compute units idle with 96-row workgroups (a lone batch: 86 of them, a ragged group) run the split-mx
    family's 64-row or 32-row form (ccsm_api.hip::launch_run picks by rounds of 256 workgroups).  A row's arithmetic does not depend on
    which rows share its workgroup, so all three forms must produce the same bits - for a lone 2048-site batch, a ragged size, and a
    tiny one; CCSM_WG_TILES = 1 | 2 | 3 forces a form (read per launch)."""
    w, dm = model7
    for n in (2048, 700, 33):
        s = synth.synth_sites(n, 77 + n)
        h1, h2 = synth.synth_h0(n, 78 + n)
        ws = dm.workspace(n)
        monkeypatch.delenv("CCSM_WG_TILES", raising=False)
        la, pa = _fwd(ws, s, (h1, h2))
        la, pa = np.array(la), np.array(pa)
        for form in ("1", "2", "3"):
            monkeypatch.setenv("CCSM_WG_TILES", form)
            lb, pb = _fwd(ws, s, (h1, h2))
            assert np.array_equal(la, np.asarray(lb)) and np.array_equal(pa, np.asarray(pb)), (n, form, dm.precision)
        monkeypatch.delenv("CCSM_WG_TILES", raising=False)
        ws.close()
        err = np.abs(pa - _oracle(w, s, h1, h2)[1]).max()
        assert err < (DEFAULT_TOL if dm.precision >= 4 else SPLIT3_TOL), (n, err)


def test_layer0_staggered_and_lock_step_agree_bitwise(model7, monkeypatch):
    """96-row launches run layer 0 in its staggered form (waves 0-3 and 4-7 one slot apart, three barriers per step: DESIGN 7.3); a model
    created with CCSM_L0_LOCKSTEP=1 runs the lock-step form.  Both issue the same products in the same order: same bits, in every
    arithmetic, for a full coalesced-size launch and a ragged one (CCSM_WG_TILES=3 keeps both on 96-row workgroups)."""
    from ccsmeth_amd.models import DeviceModel
    w, dm = model7
    monkeypatch.setenv("CCSM_L0_LOCKSTEP", "1")
    dl = DeviceModel(w, device=0, precision=dm.precision)
    monkeypatch.delenv("CCSM_L0_LOCKSTEP")
    monkeypatch.setenv("CCSM_WG_TILES", "3")
    try:
        for n in (6144, 1000):
            s = synth.synth_sites(n, 177 + n)
            h1, h2 = synth.synth_h0(n, 178 + n)
            wa, wb = dm.workspace(n), dl.workspace(n)
            la, pa = _fwd(wa, s, (h1, h2))
            lb, pb = _fwd(wb, s, (h1, h2))
            wa.close(); wb.close()
            assert np.array_equal(np.asarray(la), np.asarray(lb)) and np.array_equal(np.asarray(pa), np.asarray(pb)), (n, dm.precision)
    finally:
        dl.close()


def test_split3_on_both_mfma_shapes(monkeypatch):
    """split3 runs its GRU layers on v_mfma_f32_16x16x32_f16 (ccsm_gru_f3s.hip: the shape that sustains 17 % more under the power cap); a
    model created with CCSM_F3_SHAPE32=1 keeps the 32x32x16 kernels of round 4 (same inputs, outputs and weight-stream sizes, another
    order of the k sum inside an instruction).  Both stay within split3's tolerance of the oracle and within 5e-7 of each other, for a
    coalesced-size launch, a ragged one and the small-launch forms; trained weights: tests/test_gpu_zz_trained_checkpoints.py runs on the
    default (16x16x32) kernels."""
    from ccsmeth_amd.models import DeviceModel
    w = synth.synth_weights(7)
    d16 = DeviceModel(w, device=0, precision=3)
    monkeypatch.setenv("CCSM_F3_SHAPE32", "1")
    d32 = DeviceModel(w, device=0, precision=3)
    monkeypatch.delenv("CCSM_F3_SHAPE32")
    try:
        for n, form in ((6144, None), (1000, None), (513, "2"), (64, "1"), (200, "3")):
            if form is None:
                monkeypatch.delenv("CCSM_WG_TILES", raising=False)
            else:
                monkeypatch.setenv("CCSM_WG_TILES", form)
            s = synth.synth_sites(n, 277 + n)
            h1, h2 = synth.synth_h0(n, 278 + n)
            wa, wb = d16.workspace(n), d32.workspace(n)
            _, pa = _fwd(wa, s, (h1, h2))
            _, pb = _fwd(wb, s, (h1, h2))
            wa.close(); wb.close()
            pa, pb = np.asarray(pa), np.asarray(pb)
            assert np.abs(pa - pb).max() < 5e-7, (n, form, np.abs(pa - pb).max())
            if n <= 1000:
                ref = _oracle(w, s, h1, h2)[1]
                assert np.abs(pa - ref).max() < SPLIT3_TOL and np.abs(pb - ref).max() < SPLIT3_TOL, (n, form)
    finally:
        monkeypatch.delenv("CCSM_WG_TILES", raising=False)
        d16.close(); d32.close()


def test_split_mx_on_both_mfma_shapes(monkeypatch):
    """Plain split-mx has a second form of its layers 1-2 on v_mfma_f32_16x16x32_f16 + v_mfma_scale_f32_16x16x128_f8f6f4 (ccsm_gru_mx16.hip: the
    shapes that sustain 15 % more under the power cap; built in round 6, measured 10 % SLOWER than gru_layer12_mx_kernel - both kernels sit on
    the CU's vector-memory path, not on the matrix pipe - and therefore opt-in: a model created with CCSM_MX_SHAPE16=1).  Same operand values
    (fp16 hi, the same fp4 / fp6 blobs and block scales), another order of the k sum: both within the default tolerance of the oracle and
    within 2e-6 of each other, for a coalesced-size launch, a ragged one and the small-launch forms (explicit initial states: the coarse
    first-step scale of the state blobs is exercised)."""
    from ccsmeth_amd.models import DeviceModel
    w = synth.synth_weights(7)
    d32 = DeviceModel(w, device=0, precision=4)
    monkeypatch.setenv("CCSM_MX_SHAPE16", "1")
    d16 = DeviceModel(w, device=0, precision=4)
    monkeypatch.delenv("CCSM_MX_SHAPE16")
    try:
        for n, form in ((6144, None), (1000, None), (513, "2"), (64, "1"), (200, "3"), (100, "2")):
            if form is None:
                monkeypatch.delenv("CCSM_WG_TILES", raising=False)
            else:
                monkeypatch.setenv("CCSM_WG_TILES", form)
            s = synth.synth_sites(n, 377 + n)
            h1, h2 = synth.synth_h0(n, 378 + n)
            wa, wb = d16.workspace(n), d32.workspace(n)
            la, pa = _fwd(wa, s, (h1, h2))
            lb, pb = _fwd(wb, s, (h1, h2))
            wa.close(); wb.close()
            pa, pb = np.asarray(pa), np.asarray(pb)
            assert np.isfinite(pa).all() and np.isfinite(np.asarray(la)).all()
            assert np.abs(pa - pb).max() < 2e-6, (n, form, np.abs(pa - pb).max())
            if n <= 1000:
                ref = _oracle(w, s, h1, h2)[1]
                assert np.abs(pa - ref).max() < DEFAULT_TOL and np.abs(pb - ref).max() < DEFAULT_TOL, (n, form, np.abs(pa - ref).max())
    finally:
        monkeypatch.delenv("CCSM_WG_TILES", raising=False)
        d16.close(); d32.close()


def test_input_layout_variants_agree(model7):
    """float32 k-mers and per-base npass (what the reference's FloatTensor call passes) == u8 k-mers + per-site npass."""
    w, dm = model7
    n = 77
    s = synth.synth_sites(n, 31)
    h1, h2 = synth.synth_h0(n, 32)
    ws = dm.workspace(n)
    l0, p0 = _fwd(ws, s, (h1, h2))
    s2 = dict(s)
    for i in (1, 2):
        s2[f"kmer{i}"] = s[f"kmer{i}"].astype(np.float32)
        s2[f"npass{i}"] = np.repeat(s[f"npass{i}"][:, None], 21, axis=1)
    l1, p1 = _fwd(ws, s2, (h1, h2))
    ws.close()
    assert np.array_equal(l0, l1) and np.array_equal(p0, p1)


def test_batch_composition_independence(model7):
    """A site's result does not depend on which other sites share its batch (rows are independent)."""
    w, dm = model7
    n = 2048
    s = synth.synth_sites(n, 41)
    h1, h2 = synth.synth_h0(n, 42)
    ws = dm.workspace(n)
    _, p_all = _fwd(ws, s, (h1, h2))
    sel = np.arange(100, 100 + 333)
    sub = {k: v[sel] for k, v in s.items()}
    _, p_sub = _fwd(ws, sub, (h1[:, sel], h2[:, sel]))
    ws.close()
    assert np.isfinite(p_all).all()
    assert np.abs(p_all.sum(1) - 1).max() < 1e-6
    assert np.abs(p_all[sel] - p_sub).max() < 2e-6


def test_strand_swap_symmetry(model7):
    """fc1 is the only strand-asymmetric parameter: swapping the strands' inputs AND the two halves of fc1.weight
    leaves the logits unchanged."""
    from ccsmeth_amd.models import DeviceModel
    w, dm = model7
    n = 64
    s = synth.synth_sites(n, 51)
    h1, h2 = synth.synth_h0(n, 52)
    ws = dm.workspace(n)
    l0, _ = _fwd(ws, s, (h1, h2))
    ws.close()
    w2 = dict(w)
    w2["fc1.weight"] = np.concatenate([w["fc1.weight"][:, 512:], w["fc1.weight"][:, :512]], axis=1)
    sw = {}
    for k, v in s.items():
        sw[k[:-1] + ("2" if k.endswith("1") else "1")] = v
    dm2 = DeviceModel(w2, device=0)
    ws2 = dm2.workspace(n)
    l1, _ = _fwd(ws2, sw, (h2, h1))
    dm2.close()
    assert np.abs(l0 - l1).max() < 2e-5


def test_full_batch_2048_subset_vs_oracle(model7):
    w, dm = model7
    n = 2048
    s = synth.synth_sites(n, 61)
    h1, h2 = synth.synth_h0(n, 62)
    ws = dm.workspace(n)
    _, probs = _fwd(ws, s, (h1, h2))
    ws.close()
    sel = np.r_[0:48, 1000:1048, 2000:2048]
    sub = {k: v[sel] for k, v in s.items()}
    _, rp = _oracle(w, sub, h1[:, sel], h2[:, sel])
    assert np.abs(probs[sel] - rp).max() < PROB_TOL


def test_h0_modes(model7):
    w, dm = model7
    n = 256
    s = synth.synth_sites(n, 71)
    ws = dm.workspace(n)
    z = np.zeros((6, n, 256), np.float32)
    _, p_zero = _fwd(ws, s, "zero")
    _, p_exp0 = _fwd(ws, s, (z, z))
    assert np.array_equal(p_zero, p_exp0)
    _, p_a = _fwd(ws, s, None, seed=5, offset=0)
    _, p_b = _fwd(ws, s, None, seed=5, offset=0)
    _, p_c = _fwd(ws, s, None, seed=6, offset=0)
    ws.close()
    assert np.array_equal(p_a, p_b)                       # device RNG is a pure function of (seed, offset, site)
    assert np.abs(p_a - p_c).max() > 1e-3                 # and h0 matters (reference: up to 0.27 with random weights)
    # device N(0,1): mean/var of the generated states
    import ctypes as C2
    from ccsmeth_amd import _lib
    ws2 = dm.workspace(n)
    _fwd(ws2, s, None, seed=9)
    rows_p = _lib.load().ccsm_debug_rows_padded(n)
    buf = np.empty(6 * rows_p * 256, np.float32)
    _lib.check(dm._lib.ccsm_debug_read(ws2.handle, 3, buf.ctypes.data, buf.nbytes))
    ws2.close()
    g = buf.reshape(6, rows_p, 256)[:, :2 * n]
    assert abs(g.mean()) < 5e-3 and abs(g.var() - 1) < 1e-2
    assert abs(np.mean(g ** 4) - 3) < 0.1


def test_error_paths(lib, model7):
    w, dm = model7
    from ccsmeth_amd.models import DeviceModel, ModelAttRNN
    with pytest.raises(ValueError):
        ModelAttRNN(model_type="attbilstm2s")
    with pytest.raises(lib.CcsmError) as e:
        DeviceModel(w, device=0, model_type="transencoder2s")
    assert e.value.status == lib.ERR_UNSUPPORTED
    with pytest.raises(lib.CcsmError):
        DeviceModel(w, device=0, hidden_size=128)
    ws = dm.workspace(8)
    s = synth.synth_sites(9, 1)
    with pytest.raises(lib.CcsmError) as e:
        _fwd(ws, s, "zero")
    assert e.value.status == lib.ERR_CAPACITY
    ws.close()


def test_model_mirror_checkpoint_contract():
    """ModelAttRNN mirror: strict load_state_dict, DDP 'module.' prefix retry (call_modifications.py:342-358),
    16-tensor forward returning (logits, softmax) for torch tensors."""
    import torch
    from collections import OrderedDict
    from ccsmeth_amd.models import ModelAttRNN
    w = synth.synth_weights(81)
    ddp = OrderedDict(("module." + k, torch.from_numpy(v)) for k, v in w.items())
    model = ModelAttRNN(21, 3, 2, 0, 256, is_npass=True, model_type="attbigru2s", device=0)
    try:
        model.load_state_dict(ddp)
        raise AssertionError("prefixed keys must be rejected")
    except RuntimeError:
        model.load_state_dict(OrderedDict((k[7:], v) for k, v in ddp.items()))
    model.cuda(0)
    model.eval()
    n = 50
    s = synth.synth_sites(n, 82)
    h1, h2 = synth.synth_h0(n, 83)
    ft = lambda a: torch.tensor(np.asarray(a), dtype=torch.float, device="cuda:0")  # noqa: E731
    rep = lambda a: np.repeat(np.asarray(a)[:, None], 21, 1)  # noqa: E731
    zero = ft(np.zeros(n))
    logits, probs = model(ft(s["kmer1"]), ft(rep(s["npass1"])), ft(s["ipd1"]), zero, ft(s["pw1"]), zero, zero, zero,
                          ft(s["kmer2"]), ft(rep(s["npass2"])), ft(s["ipd2"]), zero, ft(s["pw2"]), zero, zero, zero,
                          h0=(torch.from_numpy(h1), torch.from_numpy(h2)))
    assert logits.is_cuda and probs.shape == (n, 2)
    _, rp = _oracle(w, s, h1, h2)
    assert np.abs(probs.cpu().numpy() - rp).max() < PROB_TOL
    assert set(model.state_dict()) == set(w)


def test_group_coalescing_matches_single_batches(model7):
    """ccsm_group_*: batches packed back to back (ragged sizes, slices starting mid-tile) and run by one launch per
    kernel give each batch exactly the result of running it alone."""
    import torch
    w, dm = model7
    sizes = [100, 33, 257, 1]
    data = []
    for i, n in enumerate(sizes):
        s = synth.synth_sites(n, 300 + i)
        h = synth.synth_h0(n, 400 + i)
        data.append((s, h))
    ws1 = dm.workspace(max(sizes))
    singles = [_fwd(ws1, s, h)[1] for s, h in data]
    ws1.close()
    wsg = dm.workspace(sum(sizes))
    dev = torch.device("cuda:0")
    outs = []
    for s, h in data:
        t = {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
        outs.append(wsg.group_add_torch(t["kmer1"], t["ipd1"], t["pw1"], t["npass1"], t["kmer2"], t["ipd2"], t["pw2"], t["npass2"],
                                        h0=(torch.from_numpy(h[0]).to(dev), torch.from_numpy(h[1]).to(dev))))
    wsg.group_run()
    torch.cuda.synchronize()
    for single, (logits, probs) in zip(singles, outs):
        assert np.abs(probs.cpu().numpy() - single).max() < 1e-6
    # capacity errors
    from ccsmeth_amd import _lib
    s, h = data[0]
    t = {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
    for _ in range(3):
        wsg.group_add_torch(t["kmer1"], t["ipd1"], t["pw1"], t["npass1"], t["kmer2"], t["ipd2"], t["pw2"], t["npass2"], h0="zero")
    with pytest.raises(_lib.CcsmError) as e:
        wsg.group_add_torch(t["kmer1"], t["ipd1"], t["pw1"], t["npass1"], t["kmer2"], t["ipd2"], t["pw2"], t["npass2"], h0="zero")
    assert e.value.status == _lib.ERR_CAPACITY
    wsg.group_run()
    torch.cuda.synchronize()
    wsg.close()


def test_call_mods2s_pipeline_vs_reference_golden():
    """extract -> _batch_feature_list2s -> _call_mods2s (HIP model, pinned h0) -> MM/ML, against the reference's own
    outputs for the same reads (tests/golden/pipeline_golden.*): integer fields exact, probabilities within 1e-4."""
    from ccsmeth_amd import _bam2modbam as mmod
    from ccsmeth_amd import call_modifications as cm
    from ccsmeth_amd import extract_features as ef
    from ccsmeth_amd.models import ModelAttRNN
    pipe = np.load(os.path.join(GOLDEN, "pipeline_golden.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "pipeline_golden.json")))
    rows = []
    for read in meta["reads"]:
        nm = read["name"]
        rows += ef.extract_features_from_double_strand_read(nm, read["seq"], pipe[nm + "_fi"], pipe[nm + "_ri"], pipe[nm + "_fp"],
                                                            pipe[nm + "_rp"], read["fn"], read["rn"])
    fb = cm._batch_feature_list2s(rows)
    c = meta["call_mods"]
    model = ModelAttRNN(21, 3, 2, 0, 256, model_type="attbigru2s", device=0)
    model.load_state_dict(synth.synth_weights(c["weight_seed"]))
    model.cuda(0).eval()
    pred, nb = cm._call_mods2s(fb, model, c["batch_size"], 0,
                               h0_provider=lambda b, n: synth.synth_h0(n, c["h0_seed_base"] + b))
    assert nb == c["batch_num"]
    assert [p[0] for p in pred] == meta["pred_holeid"] and [p[1] for p in pred] == pipe["pred_loc"].tolist()
    got = np.array([p[2] for p in pred], np.float32)
    assert np.abs(got - pipe["pred_prob"]).max() < PROB_TOL
    # ML bytes: quantised probabilities, equal except at bucket edges (+-1 there)
    reads = {r["name"]: r for r in meta["reads"]}
    for name, exp in meta["mmml"].items():
        lp = sorted((p[1], p[2]) for p in pred if p[0] == name)
        locs, probs = zip(*lp)
        assert mmod._convert_locs_to_mmtag(locs, reads[name]["seq"]) == exp["mm"]
        ml = mmod._convert_probs_to_mltag(probs)
        assert max(abs(a - b) for a, b in zip(ml, exp["ml"])) <= 1


def test_read_pipeline_double_buffered(model7):
    """reads -> features -> pinned double-buffered submit/wait -> MM/ML (ccsmeth_amd/pipeline.py): integer fields equal a
    straight per-read computation; device RNG h0 makes the probabilities a pure function of (seed, running site index)."""
    from ccsmeth_amd import _bam2modbam as mmod
    from ccsmeth_amd import extract_features as ef
    from ccsmeth_amd.pipeline import CallModsPipeline, Read
    w, dm = model7
    rng = np.random.default_rng(99)
    reads = []
    for i, length in enumerate([900, 40, 20, 3000, 1500, 25]):
        seq = rng.choice(list("ACGT"), size=length)
        for j in range(7, length - 1, 13):
            seq[j], seq[j + 1] = "C", "G"
        codes = lambda: np.clip(rng.gamma(2.0, 14.0, size=length), 0, 255).astype(np.uint8)  # noqa: E731
        reads.append(Read("read%d" % i, "".join(seq), codes(), codes(), codes(), codes(), int(rng.integers(3, 31)),
                          int(rng.integers(3, 31)), False))
    reads.append(Read("short_kin", "ACGTACGTACGTACGTACGTACGTACGTACGT", np.zeros(5, np.uint8), np.zeros(32, np.uint8),
                      np.zeros(32, np.uint8), np.zeros(32, np.uint8), 5, 5, False))
    pipe = CallModsPipeline(dm, batch_size=128, seed=77)      # small batches: reads straddle device batches
    calls, failed = pipe.run(reads)
    calls2, _ = CallModsPipeline(dm, batch_size=128, seed=77).run(reads)
    pipe.close()
    assert len(calls) == len(reads)
    n_expected_failed = 0
    for read, c, c2 in zip(reads, calls, calls2):
        arr = ef.extract_read_arrays(read.seq, read.fi, read.ri, read.fp, read.rp)
        if arr is None or len(arr["loc"]) == 0:
            n_expected_failed += 1
            assert c.mm_flag == 0 and c.n_sites == 0
            continue
        assert np.array_equal(c.locs, arr["loc"]) and c.mm_flag == 1
        assert c.mm == mmod._convert_locs_to_mmtag(arr["loc"].tolist(), read.seq)
        assert len(c.ml) == c.n_sites and all(0 <= v <= 255 for v in c.ml)
        assert np.all((c.probs >= 0) & (c.probs <= 1))
        assert np.array_equal(c.probs, c2.probs)             # reproducible run to run
    assert failed == n_expected_failed
    # GPU-side extraction: same sites, same Philox counters -> same calls
    pipe_d = CallModsPipeline(dm, batch_size=128, seed=77, extract="device")
    calls_d, failed_d = pipe_d.run(reads)
    pipe_d.close()
    assert failed_d == failed
    for c, cd in zip(calls, calls_d):
        assert c.mm_flag == cd.mm_flag and c.mm == cd.mm and np.array_equal(c.locs, cd.locs)
        if c.n_sites:
            assert np.abs(c.probs - cd.probs).max() < 2e-6


def test_aggregate_mode_vs_reference_golden():
    """Config 5 kernel (attbigru_b11, the only real checkpoint): HIP path vs the reference's own outputs, including the
    replicated per-region torch.randn stream (all -> hp1 call order), and vs the NumPy oracle."""
    from ccsmeth_amd.call_mods_freq_bam import AggrModel, _cal_modfreq_in_aggregate_mode, _cal_mod_prob, _get_normalized_histo
    g = np.load(os.path.join(GOLDEN, "aggr_golden.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "aggr_golden.json")))
    w = dict(np.load(os.path.join(GOLDEN, "aggr_ckpt_weights.npz")))
    model = AggrModel({"module." + k: v for k, v in w.items()}, device=0, tseed=meta["seed"], stream_sites=1 << 14)
    pile = synth.synth_pileup(meta["n_pile"], meta["pileup_seed"])
    pos, hist = [], []
    for p, mls in zip(pile["pos"], pile["ml"]):
        probs = [_cal_mod_prob(int(m)) for m in mls]
        if len(probs) >= 4:
            pos.append(int(p)); hist.append(_get_normalized_histo(probs))
    model.new_region()
    out_all = np.array(_cal_modfreq_in_aggregate_mode(pos, hist, model), np.float32)
    a, b = meta["sub"]
    out_hp1 = np.array(_cal_modfreq_in_aggregate_mode(pos[a:b], hist[a:b], model), np.float32)
    assert np.abs(out_all - g["out_all"]).max() < 1e-4 and np.abs(out_hp1 - g["out_hp1"]).max() < 1e-4
    # 6-dp rounded: the bulk is bit-identical.  The MFMA kernel of round 3 carries its operands as fp16 hi + lo pairs (22 to 23 bits
    # against fp32's 24), so its values sit ~2e-7 from the exact ones instead of ~1e-7 and a few per cent more of them round the other
    # way in the 6th decimal than the reference's own fp32 values do (measured: 87 % identical, max difference 1e-6 = one unit of
    # the last decimal; the fp32 vector kernel of rounds 1-2 had 93 %).  The bound that matters is the one above.
    assert np.mean(out_all == g["out_all"]) > 0.85
    assert np.abs(out_all - g["out_all"]).max() < 1.5e-6
    # stream exhaustion is an error, not silent reuse
    from ccsmeth_amd import _lib
    model.stream_pos = (1 << 14) * 64 - 64
    with pytest.raises(_lib.CcsmError) as e:
        model.forward_raw(pos[:2], np.stack(hist[:2]))
    assert e.value.status == _lib.ERR_CAPACITY
    assert _cal_modfreq_in_aggregate_mode([], [], model) is None
    # single site / tiny regions (all-padding windows)
    model.new_region()
    one = _cal_modfreq_in_aggregate_mode(pos[:1], hist[:1], model)
    assert len(one) == 1 and 0.0 <= one[0] <= 1.0
    model.close()


@pytest.mark.parametrize("case", ["aggregate_default", "aggregate_bed_cov6", "aggregate_no_comb_discrete", "aggregate_nohap_refsites_only", "aggregate_only_close"])
def test_call_freqb_aggregate_on_gpu_vs_reference_text(case, tmp_path):
    """`call_freqb --call_mode aggregate` end to end (native pile-up -> per-region windows -> HIP aggregate model -> text)
    against what the reference's own functions wrote for the same modbam (tests/golden/make_freqb_golden.py)."""
    import test_call_freqb as tf
    import torch
    from ccsmeth_amd import call_mods_freq_bam as fb
    w = dict(np.load(os.path.join(GOLDEN, "aggr_ckpt_weights.npz")))
    ckpt = str(tmp_path / "aggr.ckpt")
    torch.save({"module." + k: torch.from_numpy(v) for k, v in w.items()}, ckpt)
    a = tf._args(case, str(tmp_path / "o"), aggre_model=ckpt)
    fb.call_mods_frequency_from_bamfile(a, log=open(os.devnull, "w"))
    got = tf._outputs(str(tmp_path / "o"), a)
    diff = total = 0
    for wch in ("all", "hp1", "hp2"):
        diff += tf._close(got[wch], tf.CASES[case][wch], a.bed)
        total += len(got[wch].splitlines())
    assert total > 1000 and diff <= 0.005 * total      # lines whose 4th / 2nd decimal rounds the other way (values within _close)


def test_call_mods_bam_to_modbam(tmp_path):
    """`python -m ccsmeth_amd call_mods` end to end on a synthetic HiFi BAM: DDP-prefixed .ckpt, @PG line, MM/ML added,
    pulse tags dropped, reads without usable kinetics written untagged, reverse-strand record handled."""
    import torch
    from collections import OrderedDict
    from ccsmeth_amd import bamio
    from ccsmeth_amd import extract_features as ef
    from ccsmeth_amd import _bam2modbam as mmod
    from ccsmeth_amd.call_mods import build_parser, call_mods
    rng = np.random.default_rng(2024)
    inp = str(tmp_path / "in.bam")
    recs = []
    with bamio.BamWriter(inp, "@HD\tVN:1.5\tSO:unknown\n", []) as w:
        for i, L in enumerate([800, 30, 2500, 1200, 15, 600]):
            seq = rng.choice(list("ACGT"), size=L)
            for j in range(9, L - 1, 17):
                seq[j], seq[j + 1] = "C", "G"
            seq = "".join(seq)
            kin = lambda: rng.integers(0, 256, L).astype(np.uint8)  # noqa: E731
            tags = [("fi", "BC", kin()), ("fp", "BC", kin()), ("ri", "BC", kin()), ("rp", "BC", kin()),
                    ("fn", "C", int(rng.integers(3, 30))), ("rn", "C", int(rng.integers(3, 30))), ("np", "C", 12),
                    ("MM", "Z", "C+m,1;"), ("ML", "BC", np.array([7], np.uint8))]
            if i == 3:
                tags[0] = ("fi", "BC", kin()[:10])          # broken kinetics -> read skipped, written untagged
            flag = 16 if i == 5 else 4                       # one reverse-strand record
            stored = bamio.BamRecord("q", flag=16, seq=seq).get_forward_sequence() if i == 5 else seq
            r = bamio.BamRecord("hole%d" % i, flag=flag, ref_id=-1, seq=stored, tags=tags)
            recs.append((r, seq))
            w.write(r)
    wts = synth.synth_weights(5)
    ckpt = str(tmp_path / "m.ckpt")
    torch.save(OrderedDict(("module." + k, torch.from_numpy(v)) for k, v in wts.items()), ckpt)
    args = build_parser().parse_args(["-i", inp, "-m", ckpt, "-o", str(tmp_path / "out"), "--batch_size", "256", "--holes_batch", "4"])
    res = call_mods(args)
    assert res["reads"] == 6 and res["output"].endswith("out.modbam.bam")
    with bamio.BamReader(res["output"]) as rd:
        assert "@PG\tID:ccsmeth\tPN:ccsmeth\tVN:0.5.0" in rd.header_text
        out = list(rd)
    assert [o.query_name for o in out] == ["hole%d" % i for i in range(6)]
    tagged = 0
    for (rin, fwd), o in zip(recs, out):
        names = [t[0] for t in o.tags]
        assert not ({"fi", "fp", "ri", "rp"} & set(names)) and names.count("MM") <= 1 and "np" in names
        assert o.seq == rin.seq and o.flag == rin.flag
        arr = None if len(rin.get_tag("fi")) != len(fwd) else ef.extract_read_arrays(fwd, rin.get_tag("fi"), rin.get_tag("ri"),
                                                                                   rin.get_tag("fp"), rin.get_tag("rp"))
        if arr is None or len(arr["loc"]) == 0:
            assert "MM" not in names and "ML" not in names
            continue
        tagged += 1
        exp_mm = mmod._convert_locs_to_mmtag(arr["loc"].tolist(), fwd)
        assert o.get_tag("MM") == "C+m?," + ",".join(map(str, exp_mm)) + ";"
        ml = o.get_tag("ML")
        assert ml.dtype == np.uint8 and len(ml) == len(arr["loc"])
    assert tagged == res["tagged"] and res["failed"] == 6 - tagged
    # default = native BAM I/O + GPU-side extraction; the pure-Python I/O path with GPU extraction must write identical
    # records, and the host-extraction path the same records with ML within one bucket
    for extra, exact in ((["--io", "python"], True), (["--io", "python", "--extract", "host"], False)):
        res_h = call_mods(build_parser().parse_args(["-i", inp, "-m", ckpt, "-o", str(tmp_path / ("o" + str(exact))), "--batch_size",
                                                     "256", "--holes_batch", "4"] + extra))
        assert res_h["tagged"] == res["tagged"] and res_h["failed"] == res["failed"]
        with bamio.BamReader(res_h["output"]) as rd:
            for o, oh in zip(out, rd):
                assert [t[0] for t in o.tags] == [t[0] for t in oh.tags]
                assert (o.query_name, o.flag, o.seq) == (oh.query_name, oh.flag, oh.seq)
                if "MM" in [t[0] for t in o.tags]:
                    assert o.get_tag("MM") == oh.get_tag("MM")
                    d = np.abs(o.get_tag("ML").astype(int) - oh.get_tag("ML").astype(int)).max()
                    assert d == 0 if exact else d <= 1
    # --holeids_ne / --holeids_e (extract_features.py:268-271): excluded reads are written untagged, in both I/O paths
    ids = str(tmp_path / "ids.txt")
    open(ids, "w").write("hole0\nhole2\n")
    for extra in ([], ["--io", "python"]):
        r_ne = call_mods(build_parser().parse_args(["-i", inp, "-m", ckpt, "-o", str(tmp_path / "ne"), "--holeids_ne", ids] + extra))
        r_e = call_mods(build_parser().parse_args(["-i", inp, "-m", ckpt, "-o", str(tmp_path / "e"), "--holeids_e", ids] + extra))
        assert r_ne["tagged"] == res["tagged"] - 2 and r_e["tagged"] == 2
        with bamio.BamReader(r_e["output"]) as rd:
            assert [o.query_name for o in rd if o.has_tag("MM")] == ["hole0", "hole2"]
    # argument checks of the reference
    bad = build_parser().parse_args(["-i", inp, "-m", ckpt, "-o", str(tmp_path / "o2"), "--seq_len", "20"])
    with pytest.raises(ValueError):
        call_mods(bad)
    with pytest.raises(ValueError):
        call_mods(build_parser().parse_args(["-i", str(tmp_path / "nope.bam"), "-m", ckpt, "-o", "x"]))


def _x0_bytes(dm, ws):
    from ccsmeth_amd import _lib
    cap = dm._lib.ccsm_debug_rows_capacity(ws.handle)
    buf = np.empty((cap // 32) * 21 * 2 * 1024, np.uint8)
    _lib.check(dm._lib.ccsm_debug_read(ws.handle, 0, buf.ctypes.data, buf.nbytes))
    return buf


def _mirror_batch(reads):
    """Host-mirror features of a list of reads, concatenated in read order (what _batch_feature_list2s would hand over)."""
    from ccsmeth_amd import extract_features as ef
    parts, counts = [], []
    for r in reads:
        arr = ef.extract_read_arrays(r[0], r[1], r[2], r[3], r[4])
        counts.append(len(arr["loc"]))
        if counts[-1]:
            parts.append((arr, r[5], r[6]))
    cat = lambda key: np.concatenate([a[key] for a, _, _ in parts])  # noqa: E731
    s = dict(kmer1=cat("fkmer"), ipd1=cat("fipd").astype(np.float32), pw1=cat("fpw").astype(np.float32),
             kmer2=cat("rkmer"), ipd2=cat("ripd").astype(np.float32), pw2=cat("rpw").astype(np.float32),
             npass1=np.concatenate([np.full(len(a["loc"]), fn, np.float32) for a, fn, _ in parts]),
             npass2=np.concatenate([np.full(len(a["loc"]), rn, np.float32) for a, _, rn in parts]))
    return s, np.concatenate([a["loc"] for a, _, _ in parts]), np.array(counts)


def test_read_level_extraction_vs_reference_golden(model7):
    """ccsm_forward_reads_host: raw read arrays in, GPU extraction + model.  The layer-0 fragments it builds must be
    byte-identical to those packed from the REFERENCE's own extracted features (tests/golden/pipeline_golden.*), the site
    lists equal, and the probabilities therefore equal to the feature-level call's."""
    from ccsmeth_amd.utils.process_utils import seq_to_codes
    w, dm = model7
    pipe = np.load(os.path.join(GOLDEN, "pipeline_golden.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "pipeline_golden.json")))
    reads, exp_loc, exp_cnt, feats = [], [], [], {k: [] for k in ("kmer1", "ipd1", "pw1", "npass1", "kmer2", "ipd2", "pw2", "npass2")}
    for r in meta["reads"]:
        nm = r["name"]
        reads.append((r["seq"], pipe[nm + "_fi"], pipe[nm + "_ri"], pipe[nm + "_fp"], pipe[nm + "_rp"], r["fn"], r["rn"]))
        exp_cnt.append(r["n_sites"])
        if r["n_sites"] == 0:
            continue
        exp_loc.append(pipe[nm + "_loc"])
        feats["kmer1"].append(seq_to_codes(pipe[nm + "_fkmer"])); feats["kmer2"].append(seq_to_codes(pipe[nm + "_rkmer"]))
        feats["ipd1"].append(pipe[nm + "_fipd"]); feats["pw1"].append(pipe[nm + "_fpw"])
        feats["ipd2"].append(pipe[nm + "_ripd"]); feats["pw2"].append(pipe[nm + "_rpw"])
        feats["npass1"].append(np.full(r["n_sites"], r["fn"])); feats["npass2"].append(np.full(r["n_sites"], r["rn"]))
    s = {k: np.concatenate(v).astype(np.uint8 if k.startswith("kmer") else np.float32) for k, v in feats.items()}
    n = int(sum(exp_cnt))
    h1, h2 = synth.synth_h0(n, 4242)
    ws_r, ws_f = dm.workspace(256), dm.workspace(256)
    first, locs, logits, probs = ws_r.forward_reads(reads, h0=(h1, h2))
    lg_f, pr_f = _fwd(ws_f, s, (h1, h2))
    assert first.tolist() == np.concatenate([[0], np.cumsum(exp_cnt)]).tolist()
    assert np.array_equal(locs, np.concatenate(exp_loc))
    assert np.array_equal(_x0_bytes(dm, ws_r), _x0_bytes(dm, ws_f))
    assert np.array_equal(logits, lg_f) and np.array_equal(probs, pr_f)
    assert np.abs(probs - _oracle(w, s, h1, h2)[1]).max() < PROB_TOL
    ws_r.close(); ws_f.close()


def test_read_level_extraction_long_reads(model7):
    """Realistic read lengths (up to 25 kb), ambiguous bases, a constant-kinetics read (std == 0 -> zeros), reads without
    sites: device extraction vs the NumPy host mirror (itself pinned to the reference by tests/test_host_mirror.py)."""
    from ccsmeth_amd import _lib
    w, dm = model7
    rng = np.random.default_rng(2024)
    reads = []
    for i, length in enumerate([25000, 12, 21, 22, 23, 15000, 300, 8000, 64, 257, 511]):
        seq = rng.choice(list("ACGT"), size=length, p=[0.3, 0.2, 0.2, 0.3])
        if i == 5:
            seq[rng.integers(0, length, 40)] = "N"
        if i == 8:
            seq[:] = list("CG" * (length // 2))
        code = lambda: np.clip(rng.gamma(2.0, 20.0, size=length), 0, 255).astype(np.uint8)  # noqa: E731
        fi, ri, fp, rp = code(), code(), code(), code()
        if i == 6:
            fi[:] = 17                                          # std == 0
        reads.append(("".join(seq), fi, ri, fp, rp, int(rng.integers(3, 40)), int(rng.integers(3, 40))))
    s, exp_loc, counts = _mirror_batch(reads)
    n = len(exp_loc)
    assert n > 2000 and counts[1] == 0 and counts[2] == 0
    ws_r, ws_f = dm.workspace(n), dm.workspace(n)
    first, locs, logits, probs = ws_r.forward_reads(reads, h0="zero")
    lg_f, pr_f = _fwd(ws_f, s, "zero")
    assert np.array_equal(np.diff(first), counts) and np.array_equal(locs, exp_loc)
    a, b = _x0_bytes(dm, ws_r), _x0_bytes(dm, ws_f)
    # float64 variance sums are reduced in a different order than NumPy's pairwise sum: a last-ulp difference in std can
    # flip the 6th decimal of an isolated value; anything systematic would show up as thousands of differing bytes
    assert np.count_nonzero(a != b) <= 8
    assert np.abs(probs - pr_f).max() < 1e-6
    # device RNG h0: same running-site counter as the feature-level call
    _, _, _, p_rng = ws_r.forward_reads(reads, seed=5, offset=100)
    _, p_rng_f = _fwd(ws_f, s, None, seed=5, offset=100)
    assert np.abs(p_rng - p_rng_f).max() < 1e-6
    # capacity error: nothing computed, clean status
    small = dm.workspace(100)
    with pytest.raises(_lib.CcsmError) as ei:
        small.forward_reads(reads, h0="zero")
    assert ei.value.status == 5
    # a chunk without any site is legal and returns empty outputs
    f0, l0, _, p0 = small.forward_reads([reads[1], reads[2]], h0="zero")
    assert f0.tolist() == [0, 0, 0] and len(l0) == 0 and p0.shape == (0, 2)
    ws_r.close(); ws_f.close(); small.close()


def test_call_mods_two_ranks_equal_single_process(tmp_path):
    """Multi-GPU call_mods (one process per GPU, the input handed out in chunks, no data-path collective): two ranks — here both on
    cuda:0, bookkeeping over gloo — must produce, after stitching, exactly the records of the single-process run, in input
    order, with the same probabilities (the Philox counter of a site is (hash of its read's name, its position), whichever rank
    computes it), and the same index."""
    import subprocess
    import sys
    import torch
    from collections import OrderedDict
    from ccsmeth_amd import bamio
    from conftest import ROOT
    rng = np.random.default_rng(77)
    inp = str(tmp_path / "in.bam")
    with bamio.BamWriter(inp, "@HD\tVN:1.5\tSO:unknown\n", []) as w:
        for i in range(23):
            L = int(rng.integers(200, 3000))
            seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=L)
            pos = rng.integers(0, L - 1, L // 40)
            seq[pos], seq[pos + 1] = ord("C"), ord("G")
            kin = lambda: rng.integers(0, 256, L).astype(np.uint8)  # noqa: E731
            tags = [("fi", "BC", kin()), ("fp", "BC", kin()), ("ri", "BC", kin()), ("rp", "BC", kin()), ("fn", "C", 9), ("rn", "C", 11)]
            if i == 7:
                tags = tags[1:]                                     # unusable read in the middle
            w.write(bamio.BamRecord("z/%d/ccs" % i, flag=16 if i % 5 == 0 else 4, seq=seq.tobytes().decode(), tags=tags))
    ckpt = str(tmp_path / "m.ckpt")
    torch.save(OrderedDict((k, torch.from_numpy(v)) for k, v in synth.synth_weights(5).items()), ckpt)
    base = [sys.executable, "-m", "ccsmeth_amd", "call_mods", "-i", inp, "-m", ckpt, "--batch_size", "300", "--holes_batch", "3"]
    env = dict(os.environ, PYTHONPATH=ROOT)
    subprocess.run(base + ["-o", str(tmp_path / "single")], check=True, env=env, cwd=ROOT, timeout=600)
    procs = [subprocess.Popen(base + ["-o", str(tmp_path / "multi"), "--chunk_mb", "0.02"], cwd=ROOT,
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533"))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    assert not [f for f in os.listdir(tmp_path) if ".part" in f]

    def records(path):
        with bamio.BamReader(path) as rd:
            hdr = rd.header_text
            return hdr, [(r.query_name, r.flag, r.seq, [(t, ty, v.tolist() if isinstance(v, np.ndarray) else v) for t, ty, v in r.tags]) for r in rd]
    h1, r1 = records(str(tmp_path / "single.modbam.bam"))
    h2, r2 = records(str(tmp_path / "multi.modbam.bam"))
    assert len(r1) == 23 and r1 == r2
    assert h1.split("@PG")[0] == h2.split("@PG")[0]
    assert sum(1 for r in r1 if any(t == "ML" for t, _, _ in r[3])) >= 20
    assert open(str(tmp_path / "single.modbam.bam.bai"), "rb").read() == open(str(tmp_path / "multi.modbam.bam.bai"), "rb").read()


def test_keyed_initial_states_do_not_depend_on_order_or_batching(model7):
    """ccsm_reads.h0_key / ccsm_h0.site_key: with per-read keys a site's device-drawn initial states are a function of (key, position
    of the C), so the same reads in another order, split over several calls, or handed over as extracted features (host entry point
    with per-site keys) give bit-identical probabilities; without keys the counter is offset + running site index, as before."""
    import ctypes as C
    from ccsmeth_amd import _lib
    from ccsmeth_amd import extract_features as ef
    from ccsmeth_amd.pipeline import read_key
    _, dm = model7
    rng = np.random.default_rng(31)
    reads, names = [], []
    for i in range(9):
        L = int(rng.integers(300, 2500))
        seq = rng.choice(list("ACGT"), size=L)
        for j in range(12, L - 12, 23):
            seq[j], seq[j + 1] = "C", "G"
        kin = lambda: rng.integers(0, 256, L).astype(np.uint8)  # noqa: E731
        reads.append(("".join(seq), kin(), kin(), kin(), kin(), float(rng.integers(3, 30)), float(rng.integers(3, 30))))
        names.append("m84/%d/ccs" % (1000 + i))
    keys = np.array([read_key(n) for n in names], np.uint64)
    ws = dm.workspace(4096)
    first, locs, _, probs = ws.forward_reads(reads, seed=77, read_keys=keys)
    per_read = [probs[first[i]:first[i + 1]] for i in range(9)]
    # another order, in two calls
    order = rng.permutation(9)
    for part in (order[:4], order[4:]):
        f2, l2, _, p2 = ws.forward_reads([reads[i] for i in part], seed=77, read_keys=keys[part])
        for j, i in enumerate(part):
            assert np.array_equal(p2[f2[j]:f2[j + 1]], per_read[i])
            assert np.array_equal(l2[f2[j]:f2[j + 1]], locs[first[i]:first[i + 1]])
    # the same sites as extracted features through ccsm_forward_host with per-site keys (features are byte-identical: the
    # device extraction is pinned to the host mirror by test_read_level_extraction_vs_reference_golden)
    arrs = [ef.extract_read_arrays(r[0], r[1], r[2], r[3], r[4]) for r in reads]
    cat = lambda k: np.concatenate([a[k] for a in arrs])  # noqa: E731
    n = len(locs)
    feats = dict(kmer1=cat("fkmer"), ipd1=cat("fipd").astype(np.float32), pw1=cat("fpw").astype(np.float32),
                 kmer2=cat("rkmer"), ipd2=cat("ripd").astype(np.float32), pw2=cat("rpw").astype(np.float32),
                 npass1=np.concatenate([np.full(len(a["loc"]), r[5], np.float32) for a, r in zip(arrs, reads)]),
                 npass2=np.concatenate([np.full(len(a["loc"]), r[6], np.float32) for a, r in zip(arrs, reads)]))
    b = _lib.Batch()
    keep = []
    for s_, sfx in enumerate(("1", "2")):
        a4 = [np.ascontiguousarray(feats["kmer" + sfx], np.uint8), np.ascontiguousarray(feats["ipd" + sfx]), np.ascontiguousarray(feats["pw" + sfx]),
              np.ascontiguousarray(feats["npass" + sfx])]
        keep += a4
        b.strand[s_].kmer, b.strand[s_].ipd, b.strand[s_].pw, b.strand[s_].npass = (x.ctypes.data for x in a4)
    h = _lib.H0()
    skey = np.ascontiguousarray(np.repeat(keys, np.diff(first)), np.uint64)
    ssub = np.ascontiguousarray(cat("loc"), np.uint32)
    h.mode, h.seed, h.site_key, h.site_sub = _lib.H0_DEVICE_RNG, 77, skey.ctypes.data, ssub.ctypes.data
    lg, pr = np.empty((n, 2), np.float32), np.empty((n, 2), np.float32)
    _lib.check(dm._lib.ccsm_forward_host(dm.handle, ws.handle, n, C.byref(b), C.byref(h), lg.ctypes.data, pr.ctypes.data, None))
    assert np.array_equal(pr, probs)
    # other keys, other states; no keys: the running-index counter
    f3, _, _, p3 = ws.forward_reads(reads, seed=77, read_keys=keys + np.uint64(1))
    assert np.abs(p3 - probs).max() > 1e-4
    _, _, _, p4 = ws.forward_reads(reads, seed=77, offset=5)
    _, _, _, p5 = ws.forward_reads(reads, seed=77, offset=5)
    assert np.array_equal(p4, p5) and np.abs(p4 - probs).max() > 1e-4
    ws.close()


def test_read_level_submit_wait_pipelined(model7):
    """ccsm_submit_reads_host / ccsm_wait_reads_host: two chunks in flight on two workspaces and two streams, with the
    caller's site counts (no GPU round trip in submit), equal the blocking call; wrong counts are refused at wait."""
    import torch
    from ccsmeth_amd import _lib
    from ccsmeth_amd import extract_features as ef
    w, dm = model7
    rng = np.random.default_rng(31)
    reads = []
    for length in [3000, 900, 5000, 64, 1200, 2500]:
        seq = rng.choice(list("ACGT"), size=length)
        for j in range(11, length - 1, 29):
            seq[j], seq[j + 1] = "C", "G"
        code = lambda: np.clip(rng.gamma(2.0, 20.0, size=length), 0, 255).astype(np.uint8)  # noqa: E731
        reads.append(("".join(seq), code(), code(), code(), code(), 9.0, 12.0))

    def arrays(rs):
        lens = np.array([len(r[0]) for r in rs], np.int32)
        offs = np.zeros(len(rs), np.int64)
        offs[1:] = np.cumsum(lens[:-1])
        cat = [np.frombuffer("".join(r[0] for r in rs).encode(), np.uint8)] + [np.concatenate([r[k] for r in rs]) for k in range(1, 5)]
        cnt = np.array([ef.count_kept_sites(np.frombuffer(r[0].encode(), np.uint8)) for r in rs], np.int32)
        return (offs, lens, *cat, np.array([r[5] for r in rs], np.float32), np.array([r[6] for r in rs], np.float32)), cnt
    (a1, c1), (a2, c2) = arrays(reads[:3]), arrays(reads[3:])
    ws = [dm.workspace(int(max(c1.sum(), c2.sum()))) for _ in range(2)]
    streams = [torch.cuda.Stream(torch.device("cuda", 0)) for _ in range(2)]
    ws[0].submit_reads_arrays(*a1, site_counts=c1, seed=3, offset_counter=0, stream=streams[0].cuda_stream)
    ws[1].submit_reads_arrays(*a2, site_counts=c2, seed=3, offset_counter=int(c1.sum()), stream=streams[1].cuda_stream)
    out1, out2 = ws[0].wait_reads(), ws[1].wait_reads()
    ref = dm.workspace(int(c1.sum() + c2.sum()))
    f, locs, logits, probs = ref.forward_reads(reads, seed=3, offset=0)
    n1 = int(c1.sum())
    assert np.array_equal(np.diff(f), np.concatenate([c1, c2]))
    assert np.array_equal(np.concatenate([out1[1], out2[1]]), locs)
    assert np.abs(np.concatenate([out1[3], out2[3]]) - probs).max() < 2e-6
    assert out1[0][-1] == n1 and out2[0][-1] == int(c2.sum())
    # wrong counts: refused when collected, workspace usable afterwards
    bad = c1.copy()
    bad[1] -= 1
    ws[0].submit_reads_arrays(*a1, site_counts=bad, seed=3)
    with pytest.raises(_lib.CcsmError):
        ws[0].wait_reads()
    ws[0].submit_reads_arrays(*a1, seed=3)                     # no counts: one round trip inside submit
    again = ws[0].wait_reads()
    assert np.array_equal(again[1], out1[1]) and np.abs(again[3] - out1[3]).max() < 2e-6
    # a second submit before the wait is an error
    ws[1].submit_reads_arrays(*a2, site_counts=c2, seed=3)
    with pytest.raises(_lib.CcsmError):
        ws[1].submit_reads_arrays(*a2, site_counts=c2, seed=3)
    ws[1].wait_reads()
    for x in ws + [ref]:
        x.close()


def test_call_mods_post_processing_sort_and_index(tmp_path):
    """Without --no_sort the reference runs samtools sort + index on the modbam (call_modifications.py:592-607): an unaligned
    input keeps its order, gets @HD SO:coordinate and a .bai; an aligned, unsorted input comes out in coordinate order with
    its MM/ML tags, indexed; --no_sort leaves neither."""
    import torch
    from collections import OrderedDict
    from ccsmeth_amd import bamio
    from ccsmeth_amd.call_mods import build_parser, call_mods
    rng = np.random.default_rng(77)
    wts = synth.synth_weights(5)
    ckpt = str(tmp_path / "m.ckpt")
    torch.save(OrderedDict((k, torch.from_numpy(v)) for k, v in wts.items()), ckpt)

    def reads(aligned):
        out = []
        for i in range(12):
            L = int(rng.integers(300, 900))
            seq = rng.choice(list("ACGT"), size=L)
            for j in range(12, L - 12, 23):
                seq[j], seq[j + 1] = "C", "G"
            kin = lambda: rng.integers(0, 256, L).astype(np.uint8)  # noqa: E731
            tags = [("fi", "BC", kin()), ("fp", "BC", kin()), ("ri", "BC", kin()), ("rp", "BC", kin()), ("fn", "C", 9), ("rn", "C", 8)]
            if aligned:
                out.append(bamio.BamRecord("h%d" % i, flag=16 * (i % 2), ref_id=i % 2, pos=int(rng.integers(0, 5000)), mapq=60, cigar=[(0, L)],
                                           seq="".join(seq), tags=tags))
            else:
                out.append(bamio.BamRecord("h%d" % i, flag=4, seq="".join(seq), tags=tags))
        return out
    for aligned in (False, True):
        inp = str(tmp_path / ("in%d.bam" % aligned))
        refs = [("c0", 9000), ("c1", 9000)] if aligned else []
        recs = reads(aligned)
        with bamio.BamWriter(inp, "@HD\tVN:1.5\tSO:unknown\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs), refs) as w:
            for r in recs:
                w.write(r)
        for io in ("native", "python"):
            prefix = str(tmp_path / ("o%d%s" % (aligned, io)))
            res = call_mods(build_parser().parse_args(["-i", inp, "-m", ckpt, "-o", prefix, "--io", io]), log=open(os.devnull, "w"))
            assert os.path.exists(res["output"] + ".bai")
            with bamio.BamReader(res["output"]) as rd:
                out = list(rd)
                assert "SO:coordinate" in rd.header_text.split("\n")[0]
            assert all(o.has_tag("MM") and o.has_tag("ML") for o in out)
            if aligned:
                keys = [(o.ref_id, o.pos) for o in out]
                assert keys == sorted(keys) and keys != [(r.ref_id, r.pos) for r in recs]
                by_name = {o.query_name: o for o in out}
            else:
                assert [o.query_name for o in out] == [r.query_name for r in recs]
                by_name = {o.query_name: o for o in out}
            if io == "native":
                first = by_name
            else:
                for k, o in by_name.items():
                    assert o.get_tag("MM") == first[k].get_tag("MM") and np.abs(o.get_tag("ML").astype(int) - first[k].get_tag("ML").astype(int)).max() <= 1
        res = call_mods(build_parser().parse_args(["-i", inp, "-m", ckpt, "-o", str(tmp_path / ("n%d" % aligned)), "--no_sort"]), log=open(os.devnull, "w"))
        assert not os.path.exists(res["output"] + ".bai")
        with bamio.BamReader(res["output"]) as rd:
            assert [o.query_name for o in rd] == [r.query_name for r in recs] and "SO:unknown" in rd.header_text


def test_call_mods_empty_and_siteless_inputs(tmp_path):
    """Edge inputs of the CLI: a BAM with no records, and one whose reads carry no CpG that keeps a full 21-mer window."""
    import torch
    from collections import OrderedDict
    from ccsmeth_amd import bamio
    from ccsmeth_amd.call_mods import build_parser, call_mods
    ckpt = str(tmp_path / "m.ckpt")
    torch.save(OrderedDict((k, torch.from_numpy(v)) for k, v in synth.synth_weights(5).items()), ckpt)
    empty = str(tmp_path / "empty.bam")
    with bamio.BamWriter(empty, "@HD\tVN:1.5\tSO:unknown\n", []):
        pass
    siteless = str(tmp_path / "siteless.bam")
    with bamio.BamWriter(siteless, "@HD\tVN:1.5\tSO:unknown\n", []) as w:
        for i, seq in enumerate(["ATATATATATATATATATATATATATATAT", "CG", "ACGTACGTAC", "TTTTTTTTTTCGTTTTTTTTT"]):
            L = len(seq)
            k = np.arange(L, dtype=np.uint8)
            w.write(bamio.BamRecord("r%d" % i, flag=4, seq=seq, tags=[("fi", "BC", k), ("fp", "BC", k), ("ri", "BC", k), ("rp", "BC", k), ("fn", "C", 5), ("rn", "C", 5)]))
    for io in ("native", "python"):
        res = call_mods(build_parser().parse_args(["-i", empty, "-m", ckpt, "-o", str(tmp_path / ("e" + io)), "--io", io]), log=open(os.devnull, "w"))
        assert res["reads"] == 0 and res["tagged"] == 0
        with bamio.BamReader(res["output"]) as rd:
            assert list(rd) == [] and "@PG\tID:ccsmeth" in rd.header_text
        assert open(res["output"] + ".bai", "rb").read() == b"BAI\x01" + bytes(4) + bytes(8)
        res = call_mods(build_parser().parse_args(["-i", siteless, "-m", ckpt, "-o", str(tmp_path / ("s" + io)), "--io", io]), log=open(os.devnull, "w"))
        assert res["reads"] == 4 and res["tagged"] == 0
        with bamio.BamReader(res["output"]) as rd:
            out = list(rd)
        assert [o.query_name for o in out] == ["r0", "r1", "r2", "r3"] and not any(o.has_tag("MM") for o in out)


def test_config2_one_million_sites_properties():
    """BASELINE.json configs[1] at its full size (1 000 000 sites, batches of 2048, the last one 576) through size-independent
    properties: every output finite and a probability pair; with the device generator keyed by the global site index the result of
    a site does not depend on how the run is cut into batches (2048-site batches vs 8192-site batches: bit-identical); a repeated run
    is bit-identical; a checksum of the per-batch checksums equals the checksum of the whole; and over all 1 000 000 sites the default
    (split-mx) and the hybrid arithmetic stay within 1e-5 of the fp32-class three-pass arithmetic (measured: 7.2e-6 and 4.7e-6)."""
    from ccsmeth_amd.models import DeviceModel
    n = 1_000_000
    s = synth.synth_sites(n, 20260928)
    dm = DeviceModel(synth.synth_weights(20260928), device=0)
    ws = dm.workspace(8192)

    def run(batch):
        out = np.empty((n, 2), np.float32)
        for a in range(0, n, batch):
            b = min(n, a + batch)
            sub = {k: v[a:b] for k, v in s.items()}
            _, p = ws.forward_host(sub["kmer1"], sub["ipd1"], sub["pw1"], sub["npass1"], sub["kmer2"], sub["ipd2"], sub["pw2"], sub["npass2"],
                                   h0=None, seed=1234, offset=a)
            out[a:b] = p
        return out
    p2048 = run(2048)
    assert (n - 1) // 2048 + 1 == 489 and n - 488 * 2048 == 576
    assert np.isfinite(p2048).all() and (p2048 >= 0).all() and (p2048 <= 1).all() and np.abs(p2048.sum(1) - 1).max() < 1e-6
    p8192 = run(8192)
    assert np.array_equal(p2048, p8192)
    # launches of 30000 sites: a layer output of 2.6 GB, past the 2 GiB range of one buffer descriptor (the kernels rebase theirs per
    # workgroup; a launch-wide descriptor returned zeros beyond 24960 sites)
    ws.close()
    ws = dm.workspace(30000)
    assert np.array_equal(p2048, run(30000))
    ws.close()
    ws = dm.workspace(8192)
    assert np.array_equal(p2048, run(2048))
    whole = np.float64(p2048[:, 1].astype(np.float64).sum())
    parts = sum(np.float64(p2048[a:a + 2048, 1].astype(np.float64).sum()) for a in range(0, n, 2048))
    assert abs(whole - parts) < 1e-6 * whole
    assert 0.05 < p2048[:, 1].mean() < 0.95 and p2048[:, 1].std() > 1e-3          # not a constant: random weights, real variation
    ws.close()
    assert dm.precision == 4
    dm.close()
    for prec, bound in ((3, 0.0), (5, 1e-5)):
        dm = DeviceModel(synth.synth_weights(20260928), device=0, precision=prec)
        ws = dm.workspace(30000)                                               # these two in launches past 2 GiB as well
        other = run(30000)
        ws.close()
        dm.close()
        if prec == 3:
            split3 = other
            assert np.abs(p2048 - split3).max() < 1e-5                         # split-mx against the three-pass arithmetic, every site
        else:
            assert np.abs(other - split3).max() < bound


def test_call_mods_other_normalisations_and_raw_codes(tmp_path):
    """`--norm min-mean|min-max|mad|none` and `--no_decode` (extract_features.py:181-199, 327-334) run through the NumPy mirror of the
    reference's extraction (the device kernels implement the default); the probabilities are those of the model on exactly those
    features: checked against the oracle with the run's own initial states replaced by zeros is not possible (device-drawn), so the
    check is on the features' side - the run's ML bytes equal a second run's (deterministic: states keyed by read name and position),
    differ from the default normalisation's, and a forward of the library on the mirror's features with the same keys reproduces
    them."""
    import torch
    from collections import OrderedDict
    from ccsmeth_amd import bamio
    from ccsmeth_amd.call_mods import build_parser, call_mods
    rng = np.random.default_rng(31)
    inp = str(tmp_path / "in.bam")
    with bamio.BamWriter(inp, "@HD\tVN:1.5\tSO:unknown\n", []) as w:
        for i, L in enumerate([700, 1500, 400]):
            seq = rng.choice(list("ACGT"), size=L)
            for j in range(11, L - 12, 23):
                seq[j], seq[j + 1] = "C", "G"
            kin = lambda: np.clip(rng.gamma(2.0, 12.0, L), 0, 255).astype(np.uint8)  # noqa: E731
            w.write(bamio.BamRecord("m/%d/ccs" % i, flag=4, seq="".join(seq), tags=[("fi", "BC", kin()), ("fp", "BC", kin()), ("ri", "BC", kin()),
                                                                                 ("rp", "BC", kin()), ("fn", "C", 9), ("rn", "C", 11)]))
    ckpt = str(tmp_path / "m.ckpt")
    torch.save(OrderedDict((k, torch.from_numpy(v)) for k, v in synth.synth_weights(5).items()), ckpt)

    def ml_of(extra, tag):
        out = call_mods(build_parser().parse_args(["-i", inp, "-m", ckpt, "-o", str(tmp_path / tag)] + extra), log=open(os.devnull, "w"))
        with bamio.BamReader(out["output"]) as rd:
            return np.concatenate([r.get_tag("ML") for r in rd if r.has_tag("ML")])
    base = ml_of([], "z")
    host = ml_of(["--extract", "host", "--io", "python"], "zh")
    assert np.abs(base.astype(int) - host.astype(int)).max() <= 1          # the default through both extraction paths (ML +-1 at bucket edges)
    seen = [base]
    for k, extra in enumerate((["--norm", "min-mean"], ["--norm", "min-max"], ["--norm", "mad"], ["--norm", "none"], ["--norm", "zscore", "--no_decode"])):
        a = ml_of(extra, "n%d" % k)
        b = ml_of(extra, "m%d" % k)
        assert a.shape == base.shape and np.array_equal(a, b), extra
        assert all(not np.array_equal(a, s) for s in seen), extra          # another normalisation, other probabilities
        seen.append(a)


def test_call_mods2s_at_the_reference_batch_size_coalesces_its_calls():
    """the class-level drop-in at the reference's default --batch_size 512 (call_modifications.py:177-226 cuts a
    hole-batch into 512-site model calls; one such call fills 11 of 256 workgroups).  The mirror hands this library's model the
    hole-batch in launches of >= 24576 sites instead: identical results (device-drawn initial states are keyed by the running site index;
    every launch form computes the same bits), the reference's batch count, and the time of the run with a launch-filling batch size."""
    import time
    from ccsmeth_amd import call_modifications as cm
    from ccsmeth_amd.models import ModelAttRNN
    n = 30000                                           # a hole-batch of 50 reads x ~600 sites
    s = synth.synth_sites(n, 321)
    info = ["chr\t%d\t+\thole%d\t%d" % (i, i // 600, i % 600) for i in range(n)]
    z = [0] * n
    cols = []
    for sfx in ("1", "2"):
        cols += [list(s["kmer" + sfx].astype(np.float64)), list(np.repeat(s["npass" + sfx][:, None], 21, 1)), list(s["ipd" + sfx].astype(np.float64)), z,
                 list(s["pw" + sfx].astype(np.float64)), z, z, z]
    fb = (info, *cols, [0] * n)

    def model():
        m = ModelAttRNN(21, 3, 2, 0, 256, model_type="attbigru2s", device=0, seed=77)
        m.load_state_dict(synth.synth_weights(5))
        return m.cuda(0).eval()
    m_ref = model()
    m_ref.coalesces_calls = False                       # the reference's cut: 59 calls of 512 sites
    t0 = time.time(); pred_cut, nb_cut = cm._call_mods2s(fb, m_ref, 512, 0); t_cut = time.time() - t0
    m_def = model()
    cm._call_mods2s(fb, m_def, 512, 0)                  # (first call: workspace allocation)
    m_def._calls = 0
    t0 = time.time(); pred, nb = cm._call_mods2s(fb, m_def, 512, 0); t_def = time.time() - t0
    m_big = model()
    cm._call_mods2s(fb, m_big, 24576, 0)
    m_big._calls = 0
    t0 = time.time(); pred_big, nb_big = cm._call_mods2s(fb, m_big, 24576, 0); t_big = time.time() - t0
    assert nb == nb_cut == 59 and nb_big == 2
    assert [(a, b) for a, b, _ in pred] == [(a, b) for a, b, _ in pred_cut]
    assert np.array_equal(np.array([p[2] for p in pred]), np.array([p[2] for p in pred_cut]))      # the same bits as the reference's cut
    assert np.array_equal(np.array([p[2] for p in pred]), np.array([p[2] for p in pred_big]))
    print("_call_mods2s over %d sites: --batch_size 512 cut as the reference %.3f s, coalesced %.3f s, --batch_size 24576 %.3f s" % (n, t_cut, t_def, t_big))
    assert t_def <= 1.15 * t_big + 0.02
