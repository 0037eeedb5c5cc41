"""GPU parity of the whole device path (frontend -> network -> epilogue) against the CPU oracle, through the C ABI.

The network arithmetic has no reference-held golden vector (Chainer absent: "parity unpinned", see oracle header);
the oracle here is the self-oracle restating Chainer's CPU semantics, with a float64 shadow for error budgeting.
"""
import os

import numpy as np
import pytest

from oracle import am_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-4  # north_star tolerance: max-abs on fp32 log-posteriors


def _mk(eng, network, layers, params, act="relu"):
    from nnacousticmodeling_amd.engine import Model
    return Model(eng, network, layers, params, act)


def _pipe(eng):
    from nnacousticmodeling_amd.engine import Pipeline
    return Pipeline(eng)


def _ft(path):
    return orc.load_kaldi_feature_transform(path)


@pytest.mark.parametrize("n", [1, 127, 300, 1024, 2500])
def test_mlp_predict_surface_b(engine, real_ft_path, n):
    """predict_folds FF path: splice(5) + ft + 4x1024 MLP + y - logsum(y)."""
    rng = np.random.default_rng(n)
    ft = _ft(real_ft_path)
    params = orc.make_mlp_params(rng, 440, 1024, 4, 1909)
    x = rng.standard_normal((n, 40)).astype(np.float32)
    want = orc.predict(orc.MLP(params, 4), x, None, False, 11, 0, ft)
    want64 = orc.predict(orc.MLP(params, 4, dtype=np.float64), x, None, False, 11, 0, ft)
    p = _pipe(engine)
    p.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"])
    p.set_members([_mk(engine, "ff", 4, params)])
    got = p.forward_ff(x)
    assert got.shape == (n, 1909) and got.dtype == np.float32
    assert np.max(np.abs(got - want)) <= TOL
    assert np.max(np.abs(got - want64)) <= TOL


def test_mlp_evaluate_surface_a_with_priors(engine, real_ft_path, real_priors):
    """evaluate FF path: per-utterance forward, y - ap, y - logsum(y) (evaluateModelForTest.py:98-112)."""
    rng = np.random.default_rng(7)
    ft = _ft(real_ft_path)
    params = orc.make_mlp_params(rng, 440, 1024, 4, 1909)
    offsets = orc.make_offsets(rng, 12, 3000)
    x = rng.standard_normal((3000, 40)).astype(np.float32)
    data = orc.evaluate_frontend(x, 5, ft)
    want = np.concatenate(orc.evaluate_posteriors(orc.MLP(params, 4), data, offsets, ap=real_priors))
    p = _pipe(engine)
    p.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"])
    p.set_members([_mk(engine, "ff", 4, params)])
    p.set_prior(real_priors)
    got = p.forward_ff(x)
    assert np.max(np.abs(got - want)) <= TOL


def test_mlp_6x2048_fmllr_ivectors(engine, real_ft_path):
    """configs[2]: 6x2048 on 440 + 100 i-vector columns (concatenated after the transform, evaluate.py:169-171)."""
    rng = np.random.default_rng(11)
    ft = _ft(real_ft_path)
    params = orc.make_mlp_params(rng, 540, 2048, 6, 1909)
    n = 700
    x = rng.standard_normal((n, 40)).astype(np.float32)
    iv = rng.standard_normal((n, 100)).astype(np.float32)
    data = orc.evaluate_frontend(x, 5, ft, iv)
    want = orc.log_normalise(orc.MLP(params, 6)(data))
    p = _pipe(engine)
    p.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"], 100)
    p.set_members([_mk(engine, "ff", 6, params)])
    got = p.forward_ff(x, iv)
    assert np.max(np.abs(got - want)) <= TOL


@pytest.mark.parametrize("act", ["sigmoid", "tanh"])
def test_mlp_other_activations_and_odd_sizes(engine, act):
    rng = np.random.default_rng(3)
    params = orc.make_mlp_params(rng, 5 * 13, 200, 2, 39)  # monophone-sized head, nothing a multiple of a tile
    x = rng.standard_normal((333, 13)).astype(np.float32)
    want = orc.predict(orc.MLP(params, 2, act), x, None, False, 5, 0, None)
    p = _pipe(engine)
    p.set_frontend(13, range(-2, 3))
    p.set_members([_mk(engine, "ff", 2, params, act)])
    got = p.forward_ff(x)
    assert np.max(np.abs(got - want)) <= TOL


def test_ff_shards_with_halo_equal_whole(engine, real_ft_path):
    rng = np.random.default_rng(21)
    ft = _ft(real_ft_path)
    params = orc.make_mlp_params(rng, 440, 256, 2, 1909)
    n = 1000
    x = rng.standard_normal((n, 40)).astype(np.float32)
    p = _pipe(engine)
    p.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"])
    p.set_members([_mk(engine, "ff", 2, params)])
    whole = p.forward_ff(x)
    parts = []
    for lo, hi in ((0, 333), (333, 640), (640, n)):
        h_lo, h_hi = min(5, lo), min(5, n - hi)
        parts.append(p.forward_ff(x[lo - h_lo:hi + h_hi], None, hi - lo, h_lo, h_hi))
    assert np.array_equal(np.concatenate(parts), whole)  # row-independent arithmetic: bit-identical


@pytest.mark.parametrize("timedelay", [0, 5])
def test_lstm_predict_surface_b(engine, real_ft_path, timedelay):
    """predict_folds recurrent path incl. time-delay edge padding (predict_folds.py:28-68); 3x512 on 40 dims."""
    rng = np.random.default_rng(31 + timedelay)
    ft = orc.centre_frame_transform(_ft(real_ft_path))
    params = orc.make_lstm_params(rng, 40, 512, 3, 1909)
    offsets = orc.make_offsets(rng, 9, 700, lo=20, hi=200)
    x = rng.standard_normal((700, 40)).astype(np.float32)
    want = orc.predict(orc.LSTM(params, 3), x, offsets, True, 1, timedelay, ft)
    p = _pipe(engine)
    p.set_frontend(40, [0], ft["addShift"], ft["rescale"])
    p.set_members([_mk(engine, "lstm", 3, params)])
    got = p.forward_lstm(x, offsets, timedelay=timedelay)
    assert np.max(np.abs(got - want)) <= TOL


def test_lstm_evaluate_surface_a(engine, real_ft_path, real_priors):
    """evaluate recurrent path: no time-delay shift, priors subtracted (evaluateModelForTest.py:52-94)."""
    rng = np.random.default_rng(41)
    ft = orc.centre_frame_transform(_ft(real_ft_path))
    params = orc.make_lstm_params(rng, 40, 96, 2, 1909)  # U = 96 is not a multiple of 64: exercises unit padding
    offsets = orc.make_offsets(rng, 20, 900, lo=5, hi=120)
    x = rng.standard_normal((900, 40)).astype(np.float32)
    data = orc.evaluate_frontend(x, 0, ft)
    want = np.concatenate(orc.evaluate_posteriors(orc.LSTM(params, 2), data, offsets, ap=real_priors, rnn=True))
    want64 = np.concatenate(orc.evaluate_posteriors(orc.LSTM(params, 2, dtype=np.float64), data, offsets, ap=real_priors.astype(np.float64), rnn=True))
    p = _pipe(engine)
    p.set_frontend(40, [0], ft["addShift"], ft["rescale"])
    p.set_members([_mk(engine, "lstm", 2, params)])
    p.set_prior(real_priors)
    got = p.forward_lstm(x, offsets)
    assert np.max(np.abs(got - want)) <= TOL
    assert np.max(np.abs(got - want64)) <= TOL


@pytest.mark.parametrize("units,layers", [(192, 2), (1024, 2)])
def test_lstm_wide_kernels_other_widths(engine, real_ft_path, units, layers):
    """> 32 active utterances at widths other than 512: U = 192 (K not a multiple of 512: the streaming step kernel) and
    U = 1024 (K = 1024 / 2048: the deep-prefetch kernel's loop runs 2 and 4 times); plus batch-composition invariance
    across the three step kernels (the halves finish on the few-rows kernel earlier than the whole)."""
    rng = np.random.default_rng(units)
    ft = orc.centre_frame_transform(_ft(real_ft_path))
    params = orc.make_lstm_params(rng, 40, units, layers, 1909)
    offsets = orc.make_offsets(rng, 56, 2400, lo=8, hi=90)
    x = rng.standard_normal((2400, 40)).astype(np.float32)
    want = orc.predict(orc.LSTM(params, layers), x, offsets, True, 1, 0, ft)
    p = _pipe(engine)
    p.set_frontend(40, [0], ft["addShift"], ft["rescale"])
    p.set_members([_mk(engine, "lstm", layers, params)])
    got = p.forward_lstm(x, offsets)
    assert np.max(np.abs(got - want)) <= TOL
    cut = 20
    a = p.forward_lstm(x[:offsets[cut]], offsets[:cut + 1])
    b = p.forward_lstm(x[offsets[cut]:], offsets[cut:] - offsets[cut])
    assert np.array_equal(np.concatenate([a, b]), got)


def test_ensemble_logit_average_and_rpl4(engine, real_ft_path, real_priors):
    """NNWithRPL (evaluate.py:35-51): (K*master + sum folds)/(2K), folds only, and the RPL4 epilogue (RPL.py:68-74)."""
    rng = np.random.default_rng(51)
    ft = _ft(real_ft_path)
    n = 400
    x = rng.standard_normal((n, 40)).astype(np.float32)
    data = orc.evaluate_frontend(x, 5, ft)
    plist = [orc.make_mlp_params(rng, 440, 256, 2, 1909) for _ in range(4)]
    rpl = {"W": (0.05 * rng.standard_normal((1, 1909))).astype(np.float32),
           "b": (0.1 * rng.standard_normal((1, 1909))).astype(np.float32),
           "lb": np.full((1, 1909), -20.0, dtype=np.float32) + rng.standard_normal((1, 1909)).astype(np.float32)}
    omodels = [orc.MLP(q, 2) for q in plist]
    gmodels = [_mk(engine, "ff", 2, q) for q in plist]
    cases = [
        (orc.NNWithRPL(omodels[0], omodels[1:], None), gmodels, True, None),
        (orc.NNWithRPL(None, omodels[1:], None), gmodels[1:], False, None),
        (orc.NNWithRPL(omodels[0], [], orc.RPL4(rpl)), gmodels[:1], True, rpl),
        (orc.NNWithRPL(omodels[0], omodels[1:], orc.RPL4(rpl)), gmodels, True, rpl),
    ]
    for omodel, members, has_master, r in cases:
        want = np.concatenate(orc.evaluate_posteriors(omodel, data, np.array([0, 150, n]), ap=real_priors))
        p = _pipe(engine)
        p.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"])
        p.set_members(members, has_master=has_master)
        if r is not None:
            p.set_rpl4(r["W"], r["b"], r["lb"])
        p.set_prior(real_priors)
        got = p.forward_ff(x)
        assert np.max(np.abs(got - want)) <= TOL


def test_fold_average_dev_mode(engine, real_ft_path):
    """predict_folds --fold-output-dev: mean of per-fold log-posteriors, renormalised (predict_folds.py:199-220)."""
    from nnacousticmodeling_amd._lib import AM_COMBINE_LOGPOST
    rng = np.random.default_rng(61)
    ft = _ft(real_ft_path)
    x = rng.standard_normal((500, 40)).astype(np.float32)
    plist = [orc.make_mlp_params(rng, 440, 128, 2, 1909) for _ in range(3)]
    want = orc.fold_average_dev([orc.MLP(q, 2) for q in plist], x, None, False, 11, 0, ft)
    p = _pipe(engine)
    p.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"])
    p.set_members([_mk(engine, "ff", 2, q) for q in plist], combine=AM_COMBINE_LOGPOST)
    got = p.forward_ff(x)
    assert np.max(np.abs(got - want)) <= TOL


def test_api_errors(engine):
    from nnacousticmodeling_amd._lib import AmError
    rng = np.random.default_rng(0)
    p = _pipe(engine)
    p.set_frontend(40, range(-5, 6))
    m = _mk(engine, "ff", 1, orc.make_mlp_params(rng, 100, 64, 1, 39))
    p.set_members([m])
    with pytest.raises(AmError):  # frontend width 440 != network input 100
        p.forward_ff(np.zeros((10, 40), dtype=np.float32))
    with pytest.raises(ValueError):
        _mk(engine, "transformer", 1, {})   # not one of the reference's get_nn networks


def test_lstm_ensembles_ivectors_and_dev_average(engine, real_ft_path, real_priors):
    """Recurrent members through NNWithRPL (evaluate.py:35-51), i-vector columns appended after the transform
    (evaluate.py:169-171) and the dev-mode log-posterior average (predict_folds.py:199-220) on the recurrent path."""
    from nnacousticmodeling_amd._lib import AM_COMBINE_LOGPOST
    rng = np.random.default_rng(71)
    ft = orc.centre_frame_transform(_ft(real_ft_path))
    n, d_iv = 260, 12
    offsets = orc.make_offsets(rng, 7, n, lo=10, hi=80)
    x = rng.standard_normal((n, 40)).astype(np.float32)
    iv = rng.standard_normal((n, d_iv)).astype(np.float32)
    plist = [orc.make_lstm_params(rng, 40 + d_iv, 64, 2, 1909) for _ in range(3)]
    omodels = [orc.LSTM(q, 2) for q in plist]
    gmodels = [_mk(engine, "lstm", 2, q) for q in plist]
    data = orc.evaluate_frontend(x, 0, ft, iv)
    # master + 2 folds, priors subtracted
    want = np.concatenate(orc.evaluate_posteriors(orc.NNWithRPL(omodels[0], omodels[1:]), data, offsets, ap=real_priors, rnn=True))
    p = _pipe(engine)
    p.set_frontend(40, [0], ft["addShift"], ft["rescale"], d_iv)
    p.set_members(gmodels, has_master=True)
    p.set_prior(real_priors)
    got = p.forward_lstm(x, offsets, ivec=iv)
    assert np.max(np.abs(got - want)) <= TOL
    # dev-mode average of per-member log-posteriors (predict semantics, timedelay 2); predict_folds concatenates the
    # i-vectors BEFORE the (absent) transform, which for ft=None is the same matrix
    xcat = np.concatenate((x, iv), axis=1)
    want = orc.fold_average_dev(omodels, xcat, offsets, True, 1, 2, None)
    p2 = _pipe(engine)
    p2.set_frontend(40 + d_iv, [0])
    p2.set_members(gmodels, combine=AM_COMBINE_LOGPOST)
    got = p2.forward_lstm(xcat, offsets, timedelay=2)
    assert np.max(np.abs(got - want)) <= TOL


def test_chunked_paths_in_a_fresh_process(real_ft_path):
    """The chunk loops (device-resident chunks and the pipelined host-to-host chunks) only engage above 2^18 / 2^16 rows;
    shrink them through the environment in a fresh process and compare with the un-chunked result."""
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from oracle import am_oracle as orc
from nnacousticmodeling_amd.engine import Engine, Model, Pipeline
rng = np.random.default_rng(3)
ft = orc.load_kaldi_feature_transform(sys.argv[1])
params = orc.make_mlp_params(rng, 440, 128, 2, 1909)
x = rng.standard_normal((1500, 40)).astype(np.float32)
eng = Engine(0)
p = Pipeline(eng); p.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"]); p.set_members([Model(eng, "ff", 2, params)])
y1 = p.forward_ff(x)                      # pipelined host-to-host chunks (AM_FF_PIPE_ROWS)
p.ff_upload(x, None, 1500, 0, 0); p.ff_run(); y2 = p.ff_download()   # device-resident chunks (AM_FF_CHUNK_ROWS)
np.save(sys.argv[2], np.stack([y1, y2]))
'''
    import tempfile
    outs = []
    for env in ({"AM_FF_CHUNK_ROWS": "512", "AM_FF_PIPE_ROWS": "256"}, {}):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "y.npy")
            res = subprocess.run([sys.executable, "-c", code, real_ft_path, out], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300,
                                 cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            assert res.returncode == 0, res.stderr[-2000:]
            outs.append(np.load(out))
    assert np.array_equal(outs[0][0], outs[0][1]) and np.array_equal(outs[0], outs[1])  # chunking never changes a row


def test_split_f16_precision_mode(engine, real_ft_path, real_priors):
    """AM_PRECISION_SPLIT_F16: f32 operands as hi+lo f16 halves on the f16 matrix pipe.  Must meet the same 1e-4 bound
    and be no worse against the float64 shadow than the exact f32 path (x2 slack); ensembles use the same epilogues."""
    rng = np.random.default_rng(81)
    ft = _ft(real_ft_path)
    n = 1200
    x = rng.standard_normal((n, 40)).astype(np.float32)
    params = orc.make_mlp_params(rng, 440, 1024, 4, 1909)
    want = orc.predict(orc.MLP(params, 4), x, None, False, 11, 0, ft)
    want64 = orc.predict(orc.MLP(params, 4, dtype=np.float64), x, None, False, 11, 0, ft)
    p = _pipe(engine)
    p.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"])
    p.set_members([_mk(engine, "ff", 4, params)])
    exact = p.forward_ff(x)
    try:
        engine.set_precision("split_f16")
        got = p.forward_ff(x)
        assert not p.overflowed()
        assert np.max(np.abs(got - want)) <= TOL
        e_split, e_exact = np.max(np.abs(got - want64)), np.max(np.abs(exact - want64))
        assert e_split <= max(2 * e_exact, 2e-5), (e_split, e_exact)
        # shards with halo still reproduce the whole file bit for bit in this mode
        a = p.forward_ff(x[:605], None, 600, 0, 5)
        b = p.forward_ff(x[595:], None, n - 600, 5, 0)
        assert np.array_equal(np.concatenate([a, b]), got)
        # ensemble + priors + RPL-free logit averaging through the f32 output epilogues
        plist = [orc.make_mlp_params(rng, 440, 256, 2, 1909) for _ in range(3)]
        data = orc.evaluate_frontend(x, 5, ft)
        want_e = np.concatenate(orc.evaluate_posteriors(orc.NNWithRPL(orc.MLP(plist[0], 2), [orc.MLP(q, 2) for q in plist[1:]]), data, np.array([0, n]), ap=real_priors))
        pe = _pipe(engine)
        pe.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"])
        pe.set_members([_mk(engine, "ff", 2, q) for q in plist], has_master=True)
        pe.set_prior(real_priors)
        assert np.max(np.abs(pe.forward_ff(x) - want_e)) <= TOL
        # an activation that cannot be carried in f16 raises the flag; the one-shot call falls back to the exact path
        big = {k: v.copy() for k, v in params.items()}
        big["layer_0/W"] *= 1e5       # first hidden layer ~1e5: beyond the f16 range ...
        big["layer_0/b"] *= 1e5
        big["layer_1/W"] *= 1e-5      # ... brought back by the next layer, so the final posteriors are ordinary
        pb = _pipe(engine)
        pb.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"])
        pb.set_members([_mk(engine, "ff", 4, big)])
        want_b = orc.predict(orc.MLP(big, 4), x, None, False, 11, 0, ft)
        pb.ff_upload(x, None, n, 0, 0)
        pb.ff_run()                # resident path: the run itself repeats the pass on the exact path
        assert pb.overflowed()     # ... and reports that it did
        res_b = pb.ff_download()
        assert np.all(np.isfinite(res_b)) and np.max(np.abs(res_b - want_b)) <= TOL
        got_b = pb.forward_ff(x)   # one-shot call: automatic exact re-run too
        assert pb.overflowed() and np.array_equal(got_b, res_b)
        # evaluate --no-lab: PosteriorRunner.map_shards() hands each shard's pipeline to the in-place decoder without a download;
        # with overflowing activations the decoded posteriors must be the exact path's
        from nnacousticmodeling_amd import networks, sharding
        from nnacousticmodeling_amd.phone_recog import PhoneRecog
        from oracle import viterbi_oracle as vo
        graph = vo.make_graph(np.random.default_rng(3), 5, 1909)
        gdict = dict(graph, numPhones=5, numBiphones=len(graph["lm"]), numTriphones=len(graph["biphoneIdxIn"]))
        offs = np.array([0, 300, 700, n], dtype=np.int64)
        runner = sharding.PosteriorRunner([0], precision="split_f16")
        try:
            spec_b = networks.get_nn("ff", 4, [1024], 1909).set_params(big)
            def on_shard(ctx):
                assert ctx.pipeline.overflowed()
                dec = ctx.decoder("viterbi", lambda eng_: PhoneRecog(eng_, gdict, -4.0, 1.0))
                got = dec.recognise_pipeline(ctx.pipeline, ctx.local_offsets)
                assert np.array_equal(ctx.download(), res_b)
                return got
            (segs,) = runner.map_shards(spec_b, x, on_shard, offsets=offs, shifts=ft["shifts"], ft=ft)
        finally:
            runner.close()
        dec2 = PhoneRecog(engine, gdict, -4.0, 1.0)
        assert dec2.recognise(res_b, offs) == segs
        dec2.close()
        engine.set_precision("f32")
        assert np.array_equal(pb.forward_ff(x), got_b) and not pb.overflowed()
        # a hidden width that the exact path pads to 384 is not eligible: the mode silently stays exact
        p384 = orc.make_mlp_params(rng, 440, 384, 1, 1909)
        engine.set_precision("split_f16")
        pq = _pipe(engine)
        pq.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"])
        pq.set_members([_mk(engine, "ff", 1, p384)])
        assert np.max(np.abs(pq.forward_ff(x) - orc.predict(orc.MLP(p384, 1), x, None, False, 11, 0, ft))) <= TOL
    finally:
        engine.set_precision("f32")
