"""GPU parity of the other recurrent cells of the reference's get_nn against the CPU oracle, through the
C ABI: gru (chainer L.GRU), mgrurelu / mgrurelur (the reference's MGRU.py) and peepholelstm (StatefulPeepholeLSTM).

MGRU's arithmetic is spelled out in the reference's own MGRU.py; GRU and the peephole cell are Chainer 3.5 links: "parity
unpinned" like the other network rows (Chainer absent).  Tolerance: the north_star 1e-4 max-abs on log-posteriors, also
against a float64 shadow of the oracle."""
import numpy as np
import pytest

from oracle import am_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _params(network, rng, in_dim, units, layers, n_out):
    if network == "peepholelstm":
        return orc.make_peephole_params(rng, in_dim, units, layers, n_out)
    return orc.make_gru_params(rng, in_dim, units, layers, n_out, reset_gate=network != "mgrurelu")


def _pipe(engine, network, layers, params, ft):
    from nnacousticmodeling_amd.engine import Model, Pipeline
    p = Pipeline(engine)
    p.set_frontend(40, [0], ft["addShift"], ft["rescale"])
    p.set_members([Model(engine, network, layers, params)])
    return p


CELLS = ["gru", "mgrurelu", "mgrurelur", "peepholelstm"]


@pytest.mark.parametrize("network", CELLS)
@pytest.mark.parametrize("timedelay", [0, 3])
def test_cell_predict_surface_b(engine, real_ft_path, network, timedelay):
    """predict_folds recurrent path (predict_folds.py:28-68); U = 72 is not a multiple of 64 (unit padding), 2 layers."""
    rng = np.random.default_rng(100 + timedelay)
    ft = orc.centre_frame_transform(orc.load_kaldi_feature_transform(real_ft_path))
    params = _params(network, rng, 40, 72, 2, 1909)
    offsets = orc.make_offsets(rng, 20, 900, lo=5, hi=120)
    x = rng.standard_normal((900, 40)).astype(np.float32)
    want = orc.predict(orc.get_nn(network, 2, params), x, offsets, True, 1, timedelay, ft)
    want64 = orc.predict(orc.get_nn(network, 2, params, dtype=np.float64), x, offsets, True, 1, timedelay, ft)
    got = _pipe(engine, network, 2, params, ft).forward_lstm(x, offsets, timedelay=timedelay)
    assert got.shape == want.shape and got.dtype == np.float32
    assert np.max(np.abs(got - want)) <= TOL
    assert np.max(np.abs(got - want64)) <= TOL


@pytest.mark.parametrize("network", CELLS)
def test_cell_evaluate_surface_a_config_width(engine, real_ft_path, real_priors, network):
    """evaluate recurrent path with priors (evaluateModelForTest.py:52-94) at the BASELINE width: 3 x 512, 40 utterances
    (3 row tiles: several row groups per launch), then batch-composition invariance (bit-exact)."""
    rng = np.random.default_rng(7)
    ft = orc.centre_frame_transform(orc.load_kaldi_feature_transform(real_ft_path))
    params = _params(network, rng, 40, 512, 3, 1909)
    offsets = orc.make_offsets(rng, 40, 3000, lo=10, hi=150)
    x = rng.standard_normal((3000, 40)).astype(np.float32)
    data = orc.evaluate_frontend(x, 0, ft)
    want = np.concatenate(orc.evaluate_posteriors(orc.get_nn(network, 3, params), data, offsets, ap=real_priors, rnn=True))
    p = _pipe(engine, network, 3, params, ft)
    p.set_prior(real_priors)
    got = p.forward_lstm(x, offsets)
    assert np.max(np.abs(got - want)) <= TOL
    cut = 13
    a = p.forward_lstm(x[:offsets[cut]], offsets[:cut + 1])
    b = p.forward_lstm(x[offsets[cut]:], offsets[cut:] - offsets[cut])
    assert np.array_equal(np.concatenate([a, b]), got)
    # repeated runs on one upload: eager, hipGraph capture, replay
    p.lstm_upload(x, None, offsets, 0)
    for _ in range(3):
        p.lstm_run()
        assert np.array_equal(p.lstm_download(), got)


def test_cell_ensemble_with_lstm_members(engine, real_ft_path):
    """NNWithRPL-style logit averaging over members of different cell types (evaluate.py:35-51 does not care what the
    members are): master gru + folds lstm and mgrurelu."""
    from nnacousticmodeling_amd.engine import Model, Pipeline
    rng = np.random.default_rng(11)
    ft = orc.centre_frame_transform(orc.load_kaldi_feature_transform(real_ft_path))
    nets = [("gru", orc.make_gru_params(rng, 40, 64, 1, 1909)), ("lstm", orc.make_lstm_params(rng, 40, 64, 1, 1909)),
            ("mgrurelu", orc.make_gru_params(rng, 40, 64, 1, 1909, reset_gate=False))]
    offsets = orc.make_offsets(rng, 6, 300, lo=10, hi=90)
    x = rng.standard_normal((300, 40)).astype(np.float32)
    members = [orc.get_nn(n, 1, p) for n, p in nets]
    want = orc.predict(orc.NNWithRPL(members[0], members[1:], None), x, offsets, True, 1, 0, ft)
    p = Pipeline(engine)
    p.set_frontend(40, [0], ft["addShift"], ft["rescale"])
    p.set_members([Model(engine, n, 1, q) for n, q in nets], has_master=True)
    got = p.forward_lstm(x, offsets)
    assert np.max(np.abs(got - want)) <= TOL


def test_cell_argument_errors(engine):
    from nnacousticmodeling_amd.engine import Model
    rng = np.random.default_rng(1)
    params = orc.make_gru_params(rng, 40, 32, 1, 10)
    bad = dict(params)
    bad["layer_0/U_z/W"] = np.zeros((32, 31), np.float32)
    with pytest.raises(ValueError):
        Model(engine, "gru", 1, bad)
    with pytest.raises(KeyError):
        Model(engine, "gru", 1, orc.make_gru_params(rng, 40, 32, 1, 10, reset_gate=False))


@pytest.mark.parametrize("network,layers,units,n_utts", [("gru", 3, 512, 192), ("mgrurelu", 3, 512, 192), ("mgrurelur", 2, 512, 100), ("peepholelstm", 3, 512, 150),
                                                         ("gru", 2, 72, 40), ("mgrurelu", 1, 256, 70), ("peepholelstm", 2, 100, 33), ("gru", 4, 256, 192)])
def test_persistent_cell_form_equals_the_per_tick_kernels(engine, real_ft_path, monkeypatch, network, layers, units, n_utts):
    """Round 4: GRU / MGRU / peephole stacks as ONE persistent launch (rec_stream_kernel: roles = phases of layers, row groups by tile round-robin,
    weights streamed) against the per-tick phase kernels (AM_REC_STREAM=0): same summation order and cell arithmetic, so the same bits -- whole file,
    with a time delay, thin tails (utterances of 3..140 frames), widths with pad units; plus the oracle on a sample."""
    ft = orc.centre_frame_transform(orc.load_kaldi_feature_transform(real_ft_path))
    rng = np.random.default_rng(31 * layers + units + n_utts)
    params = _params(network, rng, 40, units, layers, 1909)
    n = 45 * n_utts
    off = orc.make_offsets(rng, n_utts, n, lo=3, hi=140)
    x = rng.standard_normal((n, 40)).astype(np.float32)
    p = _pipe(engine, network, layers, params, ft)
    monkeypatch.setenv("AM_REC_STREAM", "2")          # the persistent form whenever it fits (read at every upload; the default is the per-tick kernels)
    monkeypatch.setenv("AM_REC_SPLIT", "0")           # every phase one role over [input | state]
    whole = p.forward_lstm(x, off, timedelay=2)
    form = p.recurrent_form()
    p.lstm_run()
    assert np.array_equal(p.lstm_download(), whole)   # idempotent: counters re-armed per launch, planes written once per launch
    monkeypatch.setenv("AM_REC_SPLIT", "2")           # upper layers: input part and state part of a phase as two roles, the accumulators handed over
    halves = p.forward_lstm(x, off, timedelay=2)
    assert p.recurrent_form() == (1, 1)
    assert np.array_equal(halves, whole)
    p.lstm_run()
    assert np.array_equal(p.lstm_download(), whole)
    monkeypatch.delenv("AM_REC_SPLIT")
    monkeypatch.setenv("AM_LSTM_PERSIST", "0")        # every persistent form off (read at upload): the per-tick kernels
    ticks = p.forward_lstm(x, off, timedelay=2)
    monkeypatch.delenv("AM_LSTM_PERSIST")
    assert p.recurrent_form() == (0, 1)
    monkeypatch.delenv("AM_REC_STREAM")
    assert np.array_equal(p.forward_lstm(x, off, timedelay=2), whole) and p.recurrent_form() == (0, 1)   # the default: per-tick
    assert form == (1, 1), form
    assert np.array_equal(ticks, whole)
    k = min(n_utts, 5)
    want = orc.predict(orc.get_nn(network, layers, params), x[:off[k]], off[:k + 1], True, 1, 2, ft)
    assert np.max(np.abs(whole[:off[k]] - want)) <= TOL
    p.close()
