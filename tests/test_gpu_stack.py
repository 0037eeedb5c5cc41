"""The single-launch form of the Linear stack (gemm_stack_kernel: a work queue over (layer, row tile, column tile) with
write-through hand-off of the hidden activations between workgroups) against the per-layer launches: the same tile
arithmetic, so the posteriors must agree bit for bit -- at ragged sizes, for every activation, for ensembles, for both
output-tile shapes -- and the bounded waits must give up and fall back instead of hanging."""
import numpy as np
import pytest

from oracle import am_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _pipe(engine, members, ft, shifts, d_iv=0, prior=None, activation="relu", master=False, combine="logits"):
    from nnacousticmodeling_amd import _lib
    from nnacousticmodeling_amd.engine import Model, Pipeline
    p = Pipeline(engine)
    p.set_frontend(40, shifts, ft["addShift"], ft["rescale"], d_iv)
    mode = _lib.AM_COMBINE_LOGPOST if combine == "logpost" else _lib.AM_COMBINE_LOGITS
    p.set_members([Model(engine, "ff", layers, params, activation) for layers, params in members], master, mode)
    if prior is not None:
        p.set_prior(prior)
    return p


def _resident(p, x, iv=None):
    p.ff_upload(x, iv, x.shape[0], 0, 0)
    p.ff_run()
    return p.ff_download()


def _stack_launches(engine, p, x, iv=None):
    engine.profile(True)
    engine.profile_reset()
    y = _resident(p, x, iv)
    prof = engine.profile_read()
    engine.profile(False)
    return y, prof


@pytest.mark.parametrize("layers,units,d_iv,n,n_out,act", [
    (4, 1024, 0, 57919, 1909, "relu"),      # the core-test-like size the form exists for
    (6, 2048, 100, 20001, 1909, "relu"),    # configs[2] shape, ragged
    (2, 512, 0, 4097, 1909, "sigmoid"),     # 17 row tiles, the last one row long
    (3, 256, 0, 777, 512, "tanh"),          # 256-wide output tiles; fewer items than workgroups (forced)
    (8, 256, 0, 5000, 1909, "relu"),
    (9, 512, 0, 9000, 1909, "relu"),        # the deepest feed-forward net the reference sweeps ("ff -l 9", scripts/papers/tsd2018/run.sh:17-19)
    (12, 256, 0, 3000, 1909, "relu"),       # the deepest stack one launch takes
    (1, 768, 0, 300, 40, "relu"),           # one hidden layer, two row tiles
])
def test_stack_equals_per_layer_launches(engine, real_ft_path, monkeypatch, layers, units, d_iv, n, n_out, act):
    rng = np.random.default_rng(n + units + layers)
    ft = orc.load_kaldi_feature_transform(real_ft_path)
    params = orc.make_mlp_params(rng, 440 + d_iv, units, layers, n_out)
    x = rng.standard_normal((n, 40)).astype(np.float32)
    iv = rng.standard_normal((n, d_iv)).astype(np.float32) if d_iv else None
    monkeypatch.setenv("AM_FF_STACK", "2")          # whatever the size (read at upload and at run)
    p = _pipe(engine, [(layers, params)], ft, ft["shifts"], d_iv, activation=act)
    y, prof = _stack_launches(engine, p, x, iv)
    assert prof.get("gemm_stack", (0, 0))[0] == 1 and "gemm_hidden" not in prof, prof
    assert np.array_equal(_resident(p, x, iv), y)   # idempotent (counters re-armed per launch)
    monkeypatch.setenv("AM_FF_STACK", "0")
    ref, prof0 = _stack_launches(engine, p, x, iv)
    assert "gemm_stack" not in prof0 and prof0["gemm_hidden"][0] >= layers, prof0
    assert np.array_equal(ref, y)
    assert np.array_equal(p.forward_ff(x, iv), y)   # the one-shot pipelined path (16,384-row chunks; AM_FF_STACK is still 0 here: per-layer launches)
    monkeypatch.setenv("AM_FF_STACK", "1")
    assert np.array_equal(p.forward_ff(x, iv), y)   # ... and in its default form: one stack launch per chunk (am_api.hip ff_run_chunks)
    rows = np.unique(np.concatenate([np.arange(0, 4), np.arange(n - 4, n), rng.integers(0, n, 24)]))
    feats = orc.apply_feature_transform(orc.prepare_batch(x, rows, 11), ft)
    if iv is not None:
        feats = np.concatenate((feats, iv[rows]), axis=1)
    want = orc.MLP(params, layers, act)(feats)
    want = want - orc.logsum(want, axis=1)
    assert np.max(np.abs(y[rows] - want)) <= TOL
    p.close()


@pytest.mark.parametrize("master,combine", [(False, "logits"), (True, "logits"), (False, "logpost")])
def test_stack_ensembles(engine, real_ft_path, real_priors, monkeypatch, master, combine):
    """Members run one stack launch each; the output layer's epilogue scales / accumulates into the shared logits."""
    rng = np.random.default_rng(5 + master)
    ft = orc.load_kaldi_feature_transform(real_ft_path)
    members = [(2, orc.make_mlp_params(rng, 440, 512, 2, 1909)), (3, orc.make_mlp_params(rng, 440, 256, 3, 1909)),
               (2, orc.make_mlp_params(rng, 440, 1024, 2, 1909))]
    x = rng.standard_normal((6007, 40)).astype(np.float32)
    monkeypatch.setenv("AM_FF_STACK", "1")          # 6007 rows = 24 row tiles: the default threshold (16) admits it
    p = _pipe(engine, members, ft, ft["shifts"], prior=real_priors, master=master, combine=combine)
    y, prof = _stack_launches(engine, p, x)
    assert prof.get("gemm_stack", (0, 0))[0] == 3, prof
    monkeypatch.setenv("AM_FF_STACK", "0")
    assert np.array_equal(_resident(p, x), y)
    p.close()


def test_stack_not_taken(engine, real_ft_path, monkeypatch):
    """Shapes outside the form stay on the per-layer launches: thirteen hidden layers, a hidden width that pads to 384."""
    rng = np.random.default_rng(9)
    ft = orc.load_kaldi_feature_transform(real_ft_path)
    x = rng.standard_normal((5000, 40)).astype(np.float32)
    monkeypatch.setenv("AM_FF_STACK", "2")
    for layers, units in ((13, 256), (2, 300)):
        params = orc.make_mlp_params(rng, 440, units, layers, 1909)
        p = _pipe(engine, [(layers, params)], ft, ft["shifts"])
        y, prof = _stack_launches(engine, p, x)
        assert "gemm_stack" not in prof and prof["gemm_hidden"][0] >= layers, prof
        assert np.array_equal(p.forward_ff(x), y)
        p.close()


@pytest.mark.expects_fallback
def test_stack_gives_up_and_falls_back(engine, real_ft_path, monkeypatch, capfd):
    """Every inter-workgroup wait of the stack kernel is bounded (2 s).  With arrival targets nobody reaches
    (AM_FF_TEST_TIMEOUT=1) the waits must time out, every workgroup must leave, and the download must come back with the
    per-layer launches' result -- the same bits -- instead of hanging; the pipeline then stays on the per-layer launches."""
    import time
    from nnacousticmodeling_amd import _lib
    rng = np.random.default_rng(33)
    ft = orc.load_kaldi_feature_transform(real_ft_path)
    params = orc.make_mlp_params(rng, 440, 256, 2, 1909)
    x = rng.standard_normal((4500, 40)).astype(np.float32)
    monkeypatch.setenv("AM_FF_STACK", "1")
    p = _pipe(engine, [(2, params)], ft, ft["shifts"])
    good = _resident(p, x)
    before = _lib.load().am_lstm_fallback_count()
    monkeypatch.setenv("AM_FF_TEST_TIMEOUT", "1")
    t0 = time.perf_counter()
    got = _resident(p, x)
    dt = time.perf_counter() - t0
    monkeypatch.delenv("AM_FF_TEST_TIMEOUT")
    assert np.array_equal(got, good)
    assert 1.0 < dt < 20.0, dt
    assert _lib.load().am_lstm_fallback_count() == before + 1
    assert "gave up" in capfd.readouterr().err
    y, prof = _stack_launches(engine, p, x)      # stays on the per-layer launches after a give-up
    assert "gemm_stack" not in prof and np.array_equal(y, good)
    p.close()


def test_stack_launches_competing_for_the_gpu(real_ft_path, monkeypatch):
    """Two engines on the same GPU, driven from two threads: their single-launch stacks compete for the CUs, so each launch runs
    with a changing number of resident workgroups and uneven timing -- the hand-off of the hidden activations must not care
    (the queue needs no co-residency).  Every pass must reproduce, bit for bit, what the same pipeline computed alone."""
    import threading
    from nnacousticmodeling_amd import _lib
    from nnacousticmodeling_amd.engine import Engine
    monkeypatch.setenv("AM_FF_STACK", "1")
    ft = orc.load_kaldi_feature_transform(real_ft_path)
    jobs = []
    for seed, (layers, units, n) in enumerate(((4, 1024, 30011), (3, 512, 9001))):
        rng = np.random.default_rng(100 + seed)
        eng = Engine(0)
        params = orc.make_mlp_params(rng, 440, units, layers, 1909)
        x = rng.standard_normal((n, 40)).astype(np.float32)
        p = _pipe(eng, [(layers, params)], ft, ft["shifts"])
        jobs.append({"eng": eng, "p": p, "x": x, "alone": _resident(p, x), "bad": 0})
    before = _lib.load().am_lstm_fallback_count()

    def work(job, reps):
        for _ in range(reps):
            if not np.array_equal(_resident(job["p"], job["x"]), job["alone"]):
                job["bad"] += 1

    threads = [threading.Thread(target=work, args=(jobs[0], 12)), threading.Thread(target=work, args=(jobs[1], 40))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert [j["bad"] for j in jobs] == [0, 0]
    assert _lib.load().am_lstm_fallback_count() == before
    for j in jobs:
        j["p"].close()
