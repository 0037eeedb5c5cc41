"""Non-finite frames: the device path must report what the reference reports.

Chainer's ``F.relu`` is ``numpy.maximum(x, 0)`` and ``F.maximum`` / ``F.minimum`` are ``numpy.maximum`` / ``numpy.minimum``
(chainer_networks.py:19-22, RPL.py:68-74): a NaN operand gives NaN, so one NaN feature value makes every posterior row whose
context window holds it all-NaN (and, on the recurrent path, every later frame of its utterance).  IEEE ``fmaxf`` / ``fminf``
return the other operand instead and would turn those rows into plausible-looking distributions.  The oracle uses numpy's own
``np.maximum`` / ``np.minimum``; these tests assert its non-finite row pattern exactly, and that every other row keeps the
bits it has without the bad frames.  ``kw_utils.logsum``'s NaN -> -1e20 rule (kw_utils.py:38-43) applies at the end as before.
"""
import os
import warnings

import numpy as np
import pytest

from oracle import am_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _pipe(engine, ft, shifts=None):
    from nnacousticmodeling_amd.engine import Pipeline
    p = Pipeline(engine)
    p.set_frontend(40, ft["shifts"] if shifts is None else shifts, ft["addShift"], ft["rescale"])
    return p


def _same_nonfinite(got, want):
    """Identical classes element by element: NaN where NaN, +inf where +inf, -inf where -inf."""
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(np.isposinf(got), np.isposinf(want))
    assert np.array_equal(np.isneginf(got), np.isneginf(want))


def _poison(x):
    """One NaN value, one +inf value, one frame of +inf, one -inf value in the last frame (whose row the splice's clamp repeats)."""
    xd = x.copy()
    xd[20, 7] = np.nan
    xd[700, 3] = np.inf
    xd[1200, :] = np.inf
    xd[-1, 5] = -np.inf
    return xd


@pytest.mark.parametrize("stack", ["1", "0"])
@pytest.mark.parametrize("with_rpl", [False, True])
def test_ff_4x1024_nan_and_inf_frames(engine, real_ft_path, real_priors, with_rpl, stack, monkeypatch):
    """4x1024 ReLU net, 2,000 frames (8 row tiles), both Linear-layer forms: the rows whose +-5 window holds a bad value are NaN
    exactly where the oracle's are, all other rows are bit-identical to the clean run."""
    from nnacousticmodeling_amd.engine import Model
    monkeypatch.setenv("AM_FF_STACK", stack)
    rng = np.random.default_rng(404)
    ft = orc.load_kaldi_feature_transform(real_ft_path)
    params = orc.make_mlp_params(rng, 440, 1024, 4, 1909)
    n = 2000
    x = rng.standard_normal((n, 40)).astype(np.float32)
    xd = _poison(x)
    rpl = None
    if with_rpl:
        rpl = {"W": (0.05 * rng.standard_normal((1, 1909))).astype(np.float32),
               "b": (0.1 * rng.standard_normal((1, 1909))).astype(np.float32),
               "lb": np.full((1, 1909), -20.0, dtype=np.float32) + rng.standard_normal((1, 1909)).astype(np.float32)}
    omodel = orc.NNWithRPL(orc.MLP(params, 4), [], orc.RPL4(rpl) if rpl else None)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")   # numpy warns about inf - inf and NaN in exp / max: that is the point
        want = np.concatenate(orc.evaluate_posteriors(omodel, orc.evaluate_frontend(xd, 5, ft), np.array([0, 900, n]),
                                                      ap=real_priors if with_rpl else None))
    p = _pipe(engine, ft)
    p.set_members([Model(engine, "ff", 4, params)], has_master=True)
    if rpl:
        p.set_rpl4(rpl["W"], rpl["b"], rpl["lb"])
        p.set_prior(real_priors)
    clean = p.forward_ff(x)
    got = p.forward_ff(xd)
    bad_rows = ~np.isfinite(want).all(axis=1)
    expect_bad = np.zeros(n, dtype=bool)
    for f in (20, 700, 1200, n - 1):
        expect_bad[max(0, f - 5):f + 6] = True
    assert np.array_equal(bad_rows, expect_bad)          # the oracle itself: 11 rows around each bad frame (6 at the clamped end)
    _same_nonfinite(got, want)
    assert np.isnan(got[bad_rows]).all()
    assert np.array_equal(got[~bad_rows], clean[~bad_rows])   # untouched rows: the same bits as without the bad frames
    assert np.max(np.abs(got[~bad_rows] - want[~bad_rows])) <= TOL
    assert np.isfinite(clean).all()


@pytest.mark.parametrize("persist", ["1", "0"])
@pytest.mark.parametrize("timedelay", [0, 5])
def test_lstm_3x512_nan_and_inf_frames(engine, real_ft_path, timedelay, persist, monkeypatch):
    """3x512 LSTM (predict_folds' recurrent path), persistent and per-step forms.  A NaN value poisons h and c of its utterance
    from that frame on; a single +inf value only saturates gates (tanh(+-inf) = +-1, sigmoid(+-inf) = 1 / 0), so its utterance stays
    finite -- and must still match the oracle; a whole +inf frame gives inf - inf = NaN in the upward product."""
    from nnacousticmodeling_amd.engine import Model
    monkeypatch.setenv("AM_LSTM_PERSIST", persist)
    rng = np.random.default_rng(505 + timedelay)
    ft = orc.centre_frame_transform(orc.load_kaldi_feature_transform(real_ft_path))
    params = orc.make_lstm_params(rng, 40, 512, 3, 1909)
    n = 2400
    offsets = orc.make_offsets(rng, 40, n, lo=20, hi=120)
    x = rng.standard_normal((n, 40)).astype(np.float32)
    xd = x.copy()
    u_nan, u_inf, u_frame = 3, 11, 27
    f_nan = int(offsets[u_nan]) + 9
    f_inf = int(offsets[u_inf]) + 4
    f_frame = int(offsets[u_frame]) + (int(offsets[u_frame + 1]) - int(offsets[u_frame])) // 2
    xd[f_nan, 13] = np.nan
    xd[f_inf, 2] = np.inf
    xd[f_frame, :] = np.inf
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = orc.predict(orc.LSTM(params, 3), xd, offsets, True, 1, timedelay, ft)
    p = _pipe(engine, ft, shifts=[0])
    p.set_members([Model(engine, "lstm", 3, params)])
    clean = p.forward_lstm(x, offsets, timedelay=timedelay)
    got = p.forward_lstm(xd, offsets, timedelay=timedelay)
    assert p.recurrent_form() == ((1, 1) if persist == "1" else (0, 1))
    _same_nonfinite(got, want)
    bad_rows = ~np.isfinite(want).all(axis=1)
    # the oracle's pattern: with time delay d the output row of frame f is computed at step f + d, so rows from (bad frame - d) on
    # are NaN, and the last d rows of an utterance are exactly 0 (predict_folds.py:50,61) whatever went in
    expect_bad = np.zeros(n, dtype=bool)
    for u, f in ((u_nan, f_nan), (u_frame, f_frame)):
        lo, hi = int(offsets[u]), int(offsets[u + 1])
        expect_bad[max(lo, f - timedelay):hi - timedelay] = True
    assert np.array_equal(bad_rows, expect_bad)
    touched = np.zeros(n, dtype=bool)
    for u in (u_nan, u_inf, u_frame):
        touched[int(offsets[u]):int(offsets[u + 1])] = True
    assert np.array_equal(got[~touched], clean[~touched])     # other utterances: bit-identical
    ok = ~bad_rows
    assert np.max(np.abs(got[ok] - want[ok])) <= TOL          # includes the +inf utterance's finite rows
    assert np.isfinite(clean).all()


def test_rpl4_floor_and_log_softmax_propagate_nan(engine, real_ft_path, real_priors):
    """RPL4 on an ensemble (NNWithRPL, evaluate.py:35-51): F.log_softmax has no NaN patch-up and F.maximum / F.minimum propagate
    NaN (RPL.py:69-73), so a bad row stays NaN through the floor, the prior subtraction and the final logsum."""
    from nnacousticmodeling_amd.engine import Model
    rng = np.random.default_rng(606)
    ft = orc.load_kaldi_feature_transform(real_ft_path)
    n = 600
    x = rng.standard_normal((n, 40)).astype(np.float32)
    xd = x.copy()
    xd[300, 0] = np.nan
    xd[50, 39] = -np.inf
    plist = [orc.make_mlp_params(rng, 440, 256, 2, 1909) for _ in range(3)]
    rpl = {"W": (0.05 * rng.standard_normal((1, 1909))).astype(np.float32),
           "b": (0.1 * rng.standard_normal((1, 1909))).astype(np.float32),
           "lb": np.full((1, 1909), -20.0, dtype=np.float32)}
    om = [orc.MLP(q, 2) for q in plist]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = np.concatenate(orc.evaluate_posteriors(orc.NNWithRPL(om[0], om[1:], orc.RPL4(rpl)), orc.evaluate_frontend(xd, 5, ft),
                                                      np.array([0, n]), ap=real_priors))
    p = _pipe(engine, ft)
    p.set_members([Model(engine, "ff", 2, q) for q in plist], has_master=True)
    p.set_rpl4(rpl["W"], rpl["b"], rpl["lb"])
    p.set_prior(real_priors)
    got = p.forward_ff(xd)
    _same_nonfinite(got, want)
    bad_rows = ~np.isfinite(want).all(axis=1)
    assert bad_rows.sum() == 22 and np.isnan(got[bad_rows]).all()
    assert np.max(np.abs(got[~bad_rows] - want[~bad_rows])) <= TOL


@pytest.mark.parametrize("kind,units,act", [("ff", 100, "sigmoid"), ("ff", 64, "tanh"), ("ff", 300, "sigmoid"), ("lstm", 96, None), ("lstm", 64, None),
                                            ("gru", 72, None), ("mgrurelu", 40, None), ("peepholelstm", 100, None)])
def test_padded_widths_do_not_turn_inf_into_nan(engine, kind, units, act):
    """Widths that are padded on the device (hidden widths to 256 / 384, recurrent widths to 64 / 128) have pad columns whose weights are zero.  An infinite
    input value times a zero pad weight is NaN, and the next product would spread that NaN into every real column -- while the reference, which has no
    padding, stays finite wherever an infinite pre-activation only saturates a sigmoid / tanh / gate.  The pad columns are therefore forced to 0
    (gemm_epilogue, the LSTM's hoisted projection) or never stored (the other cells).  Found by tests/fuzz_gpu.py in round 4."""
    from nnacousticmodeling_amd.engine import Model, Pipeline
    rng = np.random.default_rng(len(kind) * 1000 + units)
    D, C = 40, 39
    add = (0.1 * rng.standard_normal(D)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, D).astype(np.float32)
    ft = {"shape": [D, D], "shifts": [0], "addShift": add, "rescale": scale}
    p = Pipeline(engine)
    p.set_frontend(D, [0], add, scale, 0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if kind == "ff":
            params = orc.make_mlp_params(rng, D, units, 3, C)
            x = rng.standard_normal((300, D)).astype(np.float32)
            x[100, 3] = np.inf
            x[200, 30] = -np.inf
            want = np.concatenate(orc.evaluate_posteriors(orc.MLP(params, 3, act), orc.evaluate_frontend(x, 0, ft), np.array([0, 300])))
            p.set_members([Model(engine, "ff", 3, params, act)])
            got = p.forward_ff(x)
        else:
            layers = 2
            params = (orc.make_lstm_params if kind == "lstm" else orc.make_peephole_params if kind == "peepholelstm" else orc.make_gru_params)(
                rng, D, units, layers, C, **({"reset_gate": kind == "gru"} if kind in ("gru", "mgrurelu") else {}))
            lens = np.array([40, 33, 30, 21])
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            x = rng.standard_normal((int(off[-1]), D)).astype(np.float32)
            x[45, 7] = np.inf
            x[80, 1] = -np.inf
            want = orc.predict(orc.get_nn(kind, layers, params), x, off, True, 1, 0, ft)
            p.set_members([Model(engine, kind, layers, params)])
            got = p.forward_lstm(x, off)
    _same_nonfinite(got, want)
    fin = np.isfinite(want)
    assert fin.any() and np.max(np.abs(got[fin] - want[fin])) <= TOL
    if kind != "mgrurelu" and act != "relu":
        assert np.isfinite(want).all()      # saturating non-linearities everywhere: the reference stays finite, and so must the device
    p.close()
