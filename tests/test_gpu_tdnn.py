"""GPU parity of the TDNN network (chainer_networks.py:24-42) against the CPU oracle, through the C ABI.
The convolutions run as MFMA GEMMs over channel-last activations whose A rows are windows of the previous layer
(no im2col copy); the frontend reproduces the reference's reshape of the spliced row into [channels][positions].
Chainer's Convolution2D semantics are restated, not pinned (Chainer absent)."""
import numpy as np
import pytest

from oracle import am_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _pipe(engine, params, layers, ft, act="relu"):
    from nnacousticmodeling_amd.engine import Model, Pipeline
    p = Pipeline(engine)
    p.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"])
    p.set_members([Model(engine, "tdnn", layers, params, act)])
    return p


@pytest.mark.parametrize("units,ksize,n", [([64, 96], [3, 3], 1), ([64, 96], [3, 3], 300), ([96], [5], 777),
                                            ([256, 256, 256, 256], [5, 5, 5, 5], 700), ([128, 64, 128], [5, 1, 9], 260)])
def test_tdnn_predict_surface_b(engine, real_ft_path, units, ksize, n):
    """predict_folds FF path with the network's own window (predict_folds.py:163-186): tsd2018's 5-5-5-5 shape included."""
    rng = np.random.default_rng(n)
    ft = orc.tdnn_frame_transform(orc.load_kaldi_feature_transform(real_ft_path), ksize)
    params = orc.make_tdnn_params(rng, 40, units, ksize, 1909)
    x = rng.standard_normal((n, 40)).astype(np.float32)
    winlen = len(ft["shifts"])
    want = orc.predict(orc.TDNN(params, len(units)), x, None, False, winlen, 0, ft)
    want64 = orc.predict(orc.TDNN(params, len(units), dtype=np.float64), x, None, False, winlen, 0, ft)
    got = _pipe(engine, params, len(units), ft).forward_ff(x)
    assert got.shape == (n, 1909) and got.dtype == np.float32
    assert np.max(np.abs(got - want)) <= TOL
    assert np.max(np.abs(got - want64)) <= TOL


def test_tdnn_evaluate_surface_a_chunks_shards_and_ensemble(engine, real_ft_path, real_priors):
    """evaluate path (priors), a file longer than one device chunk (15,360 frames at a 17-position window), halo shards
    bit-identical to the whole file, and an ensemble of two TDNNs with different kernels over the same window."""
    from nnacousticmodeling_amd.engine import Model, Pipeline
    rng = np.random.default_rng(3)
    ksize = [5, 5, 5, 5]
    ft = orc.tdnn_frame_transform(orc.load_kaldi_feature_transform(real_ft_path), ksize)
    params = orc.make_tdnn_params(rng, 40, [32, 32, 32, 32], ksize, 1909, bias_scale=0.3)
    n = 17000
    x = rng.standard_normal((n, 40)).astype(np.float32)
    offsets = orc.make_offsets(rng, 50, n)
    data = orc.evaluate_frontend(x, 8, ft)
    want = np.concatenate(orc.evaluate_posteriors(orc.TDNN(params, 4, "sigmoid"), data, offsets, ap=real_priors))
    p = _pipe(engine, params, 4, ft, "sigmoid")
    p.set_prior(real_priors)
    got = p.forward_ff(x)
    assert np.max(np.abs(got - want)) <= TOL
    mid = 9000
    lo = p.forward_ff(x[:mid + 8], None, mid, 0, 8)
    hi = p.forward_ff(x[mid - 8:], None, n - mid, 8, 0)
    assert np.array_equal(lo, got[:mid]) and np.array_equal(hi, got[mid:])
    # two members, different kernel sizes, same 5-position window
    ft5 = orc.tdnn_frame_transform(orc.load_kaldi_feature_transform(real_ft_path), [3, 3])
    pa, pb = orc.make_tdnn_params(rng, 40, [48, 64], [3, 3], 1909), orc.make_tdnn_params(rng, 40, [80], [5], 1909)
    xs = x[:500]
    want = orc.predict(orc.NNWithRPL(None, [orc.TDNN(pa, 2), orc.TDNN(pb, 1)], None), xs, None, False, 5, 0, ft5)
    q = Pipeline(engine)
    q.set_frontend(40, ft5["shifts"], ft5["addShift"], ft5["rescale"])
    q.set_members([Model(engine, "tdnn", 2, pa), Model(engine, "tdnn", 1, pb)])
    assert np.max(np.abs(q.forward_ff(xs) - want)) <= TOL


def test_tdnn_split_f16_precision_mode(engine, real_ft_path):
    """AM_PRECISION_SPLIT_F16 on the convolution GEMMs (windowed A rows of hi/lo half planes): same 1e-4 bound, no worse
    than 2x the exact path against the float64 shadow; a width that the split kernels do not tile (300 -> 384 columns) stays
    on the exact path by itself."""
    rng = np.random.default_rng(12)
    ksize = [5, 5, 5, 5]
    ft = orc.tdnn_frame_transform(orc.load_kaldi_feature_transform(real_ft_path), ksize)
    n = 900
    x = rng.standard_normal((n, 40)).astype(np.float32)
    for units in ([256, 256, 512, 256], [256, 300, 256, 256]):
        params = orc.make_tdnn_params(rng, 40, units, ksize, 1909)
        want = orc.predict(orc.TDNN(params, 4), x, None, False, 17, 0, ft)
        want64 = orc.predict(orc.TDNN(params, 4, dtype=np.float64), x, None, False, 17, 0, ft)
        p = _pipe(engine, params, 4, ft)
        exact = p.forward_ff(x)
        try:
            engine.set_precision("split_f16")
            got = p.forward_ff(x)
            assert not p.overflowed()
        finally:
            engine.set_precision("f32")
        assert np.max(np.abs(got - want)) <= TOL
        assert np.max(np.abs(got - want64)) <= max(2.0 * np.max(np.abs(exact - want64)), 2e-5)
        if 300 in units:
            assert np.array_equal(got, exact)          # unsupported shape: the exact path ran
        else:
            assert not np.array_equal(got, exact)      # the split kernels really ran


def test_tdnn_argument_errors(engine, real_ft_path):
    from nnacousticmodeling_amd.engine import AmError, Model, Pipeline
    rng = np.random.default_rng(1)
    params = orc.make_tdnn_params(rng, 40, [32, 32], [3, 3], 10)
    bad = dict(params)
    bad["layer_1/W"] = np.zeros((32, 31, 1, 3), np.float32)
    with pytest.raises(ValueError):
        Model(engine, "tdnn", 2, bad)
    ft = orc.load_kaldi_feature_transform(real_ft_path)   # 11 shifts: not this network's 5-position window
    p = Pipeline(engine)
    p.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"])
    p.set_members([Model(engine, "tdnn", 2, params)])
    with pytest.raises(AmError):
        p.forward_ff(np.zeros((10, 40), np.float32))
