"""GPU parity of the individual HIP stages against the golden vectors and the CPU oracle (through the C ABI)."""
import os

import numpy as np
import pytest

from oracle import am_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-4  # north_star: fp32 outputs within 1e-4 max-abs of the CPU forward


@pytest.fixture(scope="module")
def eng(engine):
    return engine


def _pipeline(eng):
    from nnacousticmodeling_amd.engine import Pipeline
    return Pipeline(eng)


def test_splice_ft_bit_exact_vs_reference_golden(eng, golden_dir, real_ft_path):
    g = np.load(os.path.join(golden_dir, "splice.npz"))
    want = np.load(os.path.join(golden_dir, "ft_apply.npz"))["applied"]
    ft = orc.load_kaldi_feature_transform(real_ft_path)
    p = _pipeline(eng)
    p.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"], 0)
    got = p.frontend_apply(g["data"])
    assert np.array_equal(got, want)
    # splice only (no transform): identical to reference splicing()
    p.set_frontend(40, range(-5, 6))
    assert np.array_equal(p.frontend_apply(g["data"]), g["spliced5"])
    p.set_frontend(40, range(-2, 3))
    assert np.array_equal(p.frontend_apply(g["data"]), g["spliced2"])
    # file shorter than the window
    p.set_frontend(40, range(-5, 6))
    assert np.array_equal(p.frontend_apply(g["tiny"]), g["tiny5"])


def test_splice_halo_shards_equal_whole_file(eng, golden_dir):
    g = np.load(os.path.join(golden_dir, "splice.npz"))
    data, whole = g["data"], g["spliced5"]
    p = _pipeline(eng)
    p.set_frontend(40, range(-5, 6))
    n = data.shape[0]
    for lo, hi in ((0, 100), (100, 201), (201, n), (3, 9), (n - 4, n)):
        h_lo, h_hi = min(5, lo), min(5, n - hi)
        got = p.frontend_apply(data[lo - h_lo:hi + h_hi], None, hi - lo, h_lo, h_hi)
        assert np.array_equal(got, whole[lo:hi]), (lo, hi)


@pytest.mark.parametrize("D,d_iv,ctx", [(40, 100, 5), (13, 7, 2), (40, 0, 0), (23, 0, 3)])
def test_splice_general_shapes_and_ivectors(eng, D, d_iv, ctx):
    rng = np.random.default_rng(5)
    n = 301
    x = rng.standard_normal((n, D)).astype(np.float32)
    iv = rng.standard_normal((n, d_iv)).astype(np.float32) if d_iv else None
    shifts = list(range(-ctx, ctx + 1))
    ft = {"addShift": rng.standard_normal(len(shifts) * D).astype(np.float32),
          "rescale": rng.uniform(0.3, 1.0, len(shifts) * D).astype(np.float32)}
    want = orc.evaluate_frontend(x, ctx, ft, iv)
    p = _pipeline(eng)
    p.set_frontend(D, shifts, ft["addShift"], ft["rescale"], d_iv)
    assert np.array_equal(p.frontend_apply(x, iv), want)


def test_lognormalise_vs_reference_golden(eng, golden_dir, real_priors):
    g = np.load(os.path.join(golden_dir, "logsum.npz"))
    got = eng.lognormalise(g["lp"])
    want = g["norm"]
    finite = np.isfinite(want)
    # rows 3 (all -inf) and 9 (+inf) exercise the NaN -> -1e20 rule: same non-finite pattern, same values
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(got[~finite & ~np.isnan(want)], want[~finite & ~np.isnan(want)])
    assert np.max(np.abs(got[finite] - want[finite])) <= 2e-5
    assert np.all(got[9, np.arange(1909) != 100] == np.float32(1e20) + g["lp"][9, np.arange(1909) != 100])
    gp = np.load(os.path.join(golden_dir, "prior_sub.npz"))
    for coef, key in ((1.0, "coef_1"), (0.5, "coef_0p5")):
        got = eng.lognormalise(gp["y"], ap=(coef * real_priors).astype(np.float32))
        assert np.max(np.abs(got - gp[key])) <= 2e-5


@pytest.mark.parametrize("c", [39, 64, 100, 1909, 2048])
def test_lognormalise_class_counts(eng, c):
    rng = np.random.default_rng(c)
    lp = (4 * rng.standard_normal((37, c))).astype(np.float32)
    assert np.max(np.abs(eng.lognormalise(lp) - orc.log_normalise(lp))) <= 2e-5


@pytest.mark.parametrize("n,k,m,act", [(300, 440, 1024, "relu"), (129, 40, 2048, None), (1, 1024, 1909, None),
                                       (517, 540, 2048, "sigmoid"), (64, 33, 70, "tanh")])
def test_linear_mfma_vs_oracle(eng, n, k, m, act):
    rng = np.random.default_rng(n + k)
    x = rng.standard_normal((n, k)).astype(np.float32)
    w = (rng.standard_normal((m, k)) / np.sqrt(k)).astype(np.float32)
    b = rng.standard_normal(m).astype(np.float32)
    got = eng.linear(x, w, b, act)
    ref64 = x.astype(np.float64) @ w.astype(np.float64).T + b
    if act:
        ref64 = orc._activation(act)(ref64)
    ref32 = x.dot(w.T) + b
    if act:
        ref32 = orc._activation(act)(ref32)
    assert np.max(np.abs(got - ref32)) <= 2e-5
    # f32 MFMA = one k-ordered fmaf chain: error vs the float64 shadow is bounded by ~1.5e-7 * sum|x||w| (K <= 1024)
    bound = 3e-7 * np.max(np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T) + 1e-6
    assert np.max(np.abs(got - ref64)) <= bound


def test_linear_identity_asymmetric(eng):
    # A = I with an asymmetric W catches a transposed C-write or a wrong K permutation exactly
    k = 96
    x = np.eye(k, dtype=np.float32)
    w = (np.arange(160 * k, dtype=np.float32).reshape(160, k) % 251) - 125
    assert np.array_equal(eng.linear(x, w), w.T)


def test_handles_may_be_destroyed_in_any_order():
    """Host bindings finalise objects in no particular order (interpreter exit): an engine destroyed before its model,
    pipeline and decoder must stay usable for their destruction (am_engine_destroy only closes it)."""
    from nnacousticmodeling_amd.engine import Engine, Model, Pipeline
    from nnacousticmodeling_amd.phone_recog import PhoneRecog
    from oracle import viterbi_oracle as vo
    rng = np.random.default_rng(0)
    eng = Engine(0)
    m = Model(eng, "ff", 1, orc.make_mlp_params(rng, 40, 32, 1, 39))
    p = Pipeline(eng)
    p.set_frontend(40, [0])
    p.set_members([m])
    y = p.forward_ff(rng.standard_normal((10, 40)).astype(np.float32))
    graph = vo.make_graph(rng, 3, 12)
    dec = PhoneRecog(eng, dict(graph, numPhones=3, numBiphones=len(graph["lm"]), numTriphones=len(graph["biphoneIdxIn"])), -1.0, 1.0)
    eng.close()          # first
    p.close()
    dec.close()
    m.close()            # last child: the engine's resources go here
    assert y.shape == (10, 39)
