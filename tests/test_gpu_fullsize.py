"""GPU parity at BASELINE.json's full sizes through size-independent properties, plus the edge cases of the domain
(empty / tiny files, zero-length utterances, more shards than frames)."""
import numpy as np
import pytest

from oracle import am_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _pipe(engine, network, layers, params, ft, shifts, d_iv=0, prior=None):
    from nnacousticmodeling_amd.engine import Model, Pipeline
    p = Pipeline(engine)
    p.set_frontend(40, shifts, ft["addShift"], ft["rescale"], d_iv)
    p.set_members([Model(engine, network, layers, params)])
    if prior is not None:
        p.set_prior(prior)
    return p


def _sample_rows(n, rng, k=96):
    edge = np.concatenate([np.arange(0, 8), np.arange(n - 8, n)])
    return np.unique(np.concatenate([edge, rng.integers(0, n, k)]))


@pytest.mark.parametrize("layers,units,d_iv,n", [(4, 1024, 0, 196608), (4, 1024, 0, 57919), (6, 2048, 100, 57919)])
def test_ff_full_size_properties(engine, real_ft_path, real_priors, layers, units, d_iv, n):
    """configs[1] (196,608 = 192 x 1024 frames and the core-test-like 57,919) and configs[2] (6x2048 + i-vectors)."""
    rng = np.random.default_rng(n + units)
    ft = orc.load_kaldi_feature_transform(real_ft_path)
    params = orc.make_mlp_params(rng, 440 + d_iv, units, layers, 1909)
    x = rng.standard_normal((n, 40)).astype(np.float32)
    iv = rng.standard_normal((n, d_iv)).astype(np.float32) if d_iv else None
    p = _pipe(engine, "ff", layers, params, ft, ft["shifts"], d_iv, real_priors)
    y = p.forward_ff(x, iv)
    assert y.shape == (n, 1909) and np.all(np.isfinite(y))
    # (1) every row is a normalised log-distribution
    lse = orc.logsum(y, axis=1)
    assert np.max(np.abs(lse)) <= 2e-5
    # (2) sampled rows (incl. the globally clamped first/last rows) against the oracle on exactly those rows
    rows = _sample_rows(n, rng)
    feats = orc.apply_feature_transform(orc.prepare_batch(x, rows, 11), ft)
    if iv is not None:
        feats = np.concatenate((feats, iv[rows]), axis=1)
    want = orc.MLP(params, layers)(feats) - real_priors
    want = want - orc.logsum(want, axis=1)
    assert np.max(np.abs(y[rows] - want)) <= TOL
    # (3) two frame shards with halo reproduce the whole file bit for bit (row-independent arithmetic)
    mid = n // 2 + 17
    lo = p.forward_ff(x[:mid + 5], None if iv is None else iv[:mid], mid, 0, 5)
    hi = p.forward_ff(x[mid - 5:], None if iv is None else iv[mid:], n - mid, 5, 0)
    assert np.array_equal(lo, y[:mid]) and np.array_equal(hi, y[mid:])
    # (4) idempotence of the device path
    assert np.array_equal(p.forward_ff(x, iv), y)


def test_lstm_config_size(engine, real_ft_path):
    """configs[3]: 3 x 512 LSTM on a quarter of the core-test-like set (48 utterances) against the full oracle, plus
    utterance-order invariance at the full 192-utterance size."""
    rng = np.random.default_rng(99)
    ft = orc.centre_frame_transform(orc.load_kaldi_feature_transform(real_ft_path))
    params = orc.make_lstm_params(rng, 40, 512, 3, 1909)
    off = orc.make_offsets(rng, 48, 14480)
    x = rng.standard_normal((14480, 40)).astype(np.float32)
    p = _pipe(engine, "lstm", 3, params, ft, [0])
    got = p.forward_lstm(x, off, timedelay=5)
    want = orc.predict(orc.LSTM(params, 3), x, off, True, 1, 5, ft)
    assert np.max(np.abs(got - want)) <= TOL
    # full size: results must not depend on which utterances share a launch -> run all 192, then each half separately
    offs = orc.make_offsets(rng, 192, 57919)
    xx = rng.standard_normal((57919, 40)).astype(np.float32)
    whole = p.forward_lstm(xx, offs)
    assert np.max(np.abs(orc.logsum(whole, axis=1))) <= 2e-5
    cut = 96
    a = p.forward_lstm(xx[:offs[cut]], offs[:cut + 1])
    b = p.forward_lstm(xx[offs[cut]:], offs[cut:] - offs[cut])
    assert np.array_equal(np.concatenate([a, b]), whole)
    # repeated runs on one upload: 1st eager, 2nd captures a hipGraph, 3rd replays it -- all identical
    p.lstm_upload(xx, None, offs, 0)
    for _ in range(3):
        p.lstm_run()
        assert np.array_equal(p.lstm_download(), whole)


def test_edge_cases(engine, real_ft_path):
    rng = np.random.default_rng(5)
    ft = orc.load_kaldi_feature_transform(real_ft_path)
    params = orc.make_mlp_params(rng, 440, 64, 2, 1909)
    p = _pipe(engine, "ff", 2, params, ft, ft["shifts"])
    # empty file
    assert p.forward_ff(np.zeros((0, 40), dtype=np.float32)).shape == (0, 1909)
    # files shorter than the context window: every row clamps on both sides
    for n in (1, 3, 7):
        x = rng.standard_normal((n, 40)).astype(np.float32)
        want = orc.predict(orc.MLP(params, 2), x, None, False, 11, 0, ft)
        assert np.max(np.abs(p.forward_ff(x) - want)) <= TOL
    # recurrent: zero-length utterances in the middle and at the end, single-frame utterances
    ftc = orc.centre_frame_transform(ft)
    lp = orc.make_lstm_params(rng, 40, 64, 2, 1909)
    pl = _pipe(engine, "lstm", 2, lp, ftc, [0])
    off = np.array([0, 5, 5, 6, 30, 30], dtype=np.int64)
    x = rng.standard_normal((30, 40)).astype(np.float32)
    data = orc.evaluate_frontend(x, 0, ftc)
    want = np.concatenate(orc.evaluate_posteriors(orc.LSTM(lp, 2), data, off, rnn=True))
    assert np.max(np.abs(pl.forward_lstm(x, off) - want)) <= TOL
    assert pl.forward_lstm(np.zeros((0, 40), dtype=np.float32), np.array([0, 0], dtype=np.int64)).shape == (0, 1909)
    # timedelay longer than an utterance: that utterance's outputs are all zero (reference zero-tail rule)
    # (the reference itself cannot edge-pad an empty utterance -- np.pad raises -- so no empty ones here)
    off2 = np.array([0, 2, 6, 30], dtype=np.int64)
    got = pl.forward_lstm(x, off2, timedelay=3)
    want = orc.predict(orc.LSTM(lp, 2), x, off2, True, 1, 3, ftc)
    assert np.max(np.abs(got - want)) <= TOL and np.all(got[0:2] == 0) and np.all(got[3:6] == 0)


def test_multi_device_runner_on_one_gpu(engine, real_ft_path):
    """The sharded host path (threads + host gather) with two logical shards mapped onto device 0."""
    from nnacousticmodeling_amd import networks, sharding
    rng = np.random.default_rng(8)
    ft = orc.load_kaldi_feature_transform(real_ft_path)
    params = orc.make_mlp_params(rng, 440, 128, 2, 1909)
    spec = networks.get_nn("ff", 2, [128], 1909).set_params(params)
    x = rng.standard_normal((2000, 40)).astype(np.float32)
    one = sharding.PosteriorRunner([0]).forward_ff(spec, x, shifts=range(-5, 6), ft=ft)
    want = orc.predict(orc.MLP(params, 2), x, None, False, 11, 0, ft)
    assert np.max(np.abs(one - want)) <= TOL
