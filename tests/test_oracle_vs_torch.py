"""The oracle's network arithmetic cross-checked against PyTorch's own CPU implementations (independent third-party code).

Chainer, where the reference's network arithmetic lives, is absent, so the oracle's network rows are "parity unpinned"
against the reference.  What can be checked here is that the restatement computes the textbook operators it claims to:
  * L.LSTM stack  == torch.nn.LSTM after mapping Chainer's interleaved gate columns (4u + {a,i,f,o}) and bias-free lateral
    Linear onto torch's concatenated (i,f,g,o) layout;
  * TDNN          == torch conv2d on x.reshape(B, -1, 1, W), the literal expression of chainer_networks.py:39-42;
  * MLP           == torch linear + relu.
GRU / MGRU / peephole have no torch counterpart with the same equations (torch's GRU applies the reset gate after the
state Linear; Chainer's and MGRU.py's before): they are checked against a scalar loop transcription instead."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import am_oracle as orc  # noqa: E402


def test_lstm_stack_matches_torch_lstm():
    rng = np.random.default_rng(0)
    U, D, L, T, B = 24, 10, 2, 9, 3
    params = orc.make_lstm_params(rng, D, U, L, 7)
    net = torch.nn.LSTM(D, U, L)
    order = [1, 2, 0, 3]   # torch rows: i, f, g(=a), o  <- Chainer interleaved columns a, i, f, o
    with torch.no_grad():
        for l in range(L):
            up = params["layer_%d/upward/W" % l].reshape(U, 4, -1)
            ub = params["layer_%d/upward/b" % l].reshape(U, 4)
            lat = params["layer_%d/lateral/W" % l].reshape(U, 4, U)
            getattr(net, "weight_ih_l%d" % l).copy_(torch.from_numpy(np.concatenate([up[:, g, :] for g in order])))
            getattr(net, "weight_hh_l%d" % l).copy_(torch.from_numpy(np.concatenate([lat[:, g, :] for g in order])))
            getattr(net, "bias_ih_l%d" % l).copy_(torch.from_numpy(np.concatenate([ub[:, g] for g in order])))
            getattr(net, "bias_hh_l%d" % l).zero_()
    x = rng.standard_normal((T, B, D)).astype(np.float32)
    with torch.no_grad():
        h, _ = net(torch.from_numpy(x))
        want = (h @ torch.from_numpy(params["out/W"]).T + torch.from_numpy(params["out/b"])).numpy()
    model = orc.LSTM(params, L)
    model.reset_state()
    got = np.stack([model(x[t]) for t in range(T)])
    assert np.max(np.abs(got - want)) < 2e-6


def test_tdnn_matches_torch_conv2d_on_the_reference_reshape():
    rng = np.random.default_rng(1)
    ksize, units, D, B = [3, 5, 1], [6, 4, 5], 8, 11
    params = orc.make_tdnn_params(rng, D, units, ksize, 9)
    W = sum(ksize) - len(ksize) + 1
    x = rng.standard_normal((B, W * D)).astype(np.float32)
    h = torch.from_numpy(x).reshape(B, -1, 1, W)                        # chainer_networks.py:39
    for l in range(len(units)):
        h = torch.relu(torch.nn.functional.conv2d(h, torch.from_numpy(params["layer_%d/W" % l]), torch.from_numpy(params["layer_%d/b" % l])))
    want = (h.reshape(B, -1) @ torch.from_numpy(params["out/W"]).T + torch.from_numpy(params["out/b"])).numpy()
    got = orc.TDNN(params, len(units))(x)
    assert np.max(np.abs(got - want)) < 2e-6


def test_mlp_matches_torch_linear():
    rng = np.random.default_rng(2)
    params = orc.make_mlp_params(rng, 20, 16, 3, 5)
    x = rng.standard_normal((7, 20)).astype(np.float32)
    h = torch.from_numpy(x)
    for l in range(3):
        h = torch.relu(torch.nn.functional.linear(h, torch.from_numpy(params["layer_%d/W" % l]), torch.from_numpy(params["layer_%d/b" % l])))
    want = torch.nn.functional.linear(h, torch.from_numpy(params["out/W"]), torch.from_numpy(params["out/b"])).numpy()
    assert np.max(np.abs(orc.MLP(params, 3)(x) - want)) < 2e-6


@pytest.mark.parametrize("reset_gate,act", [(True, "tanh"), (True, "relu"), (False, "relu")])
def test_mgru_matches_scalar_transcription(reset_gate, act):
    """MGRU.py:67-85 transcribed element by element in float64 (no matrix calls), against the vectorised oracle."""
    rng = np.random.default_rng(3)
    U, D, T = 5, 4, 4
    p = orc.make_gru_params(rng, D, U, 1, 3, reset_gate=reset_gate)
    f = {"tanh": np.tanh, "relu": lambda v: max(v, 0.0)}[act]
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))  # noqa: E731
    lin = lambda name, vec, u: float(sum(float(p["layer_0/%s/W" % name][u, k]) * vec[k] for k in range(len(vec))) + float(p["layer_0/%s/b" % name][u]))  # noqa: E731
    x = rng.standard_normal((T, 1, D)).astype(np.float32)
    model = orc.MGRUNet(p, 1, reset_gate, act, dtype=np.float64)
    h = None
    for t in range(T):
        xt = [float(v) for v in x[t, 0]]
        new = []
        for u in range(U):
            z, hb = lin("W_z", xt, u), lin("W", xt, u)
            if h is not None:
                z += lin("U_z", h, u)
                if reset_gate:
                    rh = [sig(lin("W_r", xt, k) + lin("U_r", h, k)) * h[k] for k in range(U)]
                    hb += lin("U", rh, u)
                else:
                    hb += lin("U", h, u)
            z, hb = sig(z), f(hb)
            new.append(z * hb + (1.0 - z) * h[u] if h is not None else z * hb)
        h = new
        y = model(x[t].astype(np.float64))
        assert np.max(np.abs(model.h[0][0] - np.asarray(h))) < 1e-12
    assert y.shape == (1, 3)
